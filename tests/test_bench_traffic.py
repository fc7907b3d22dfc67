"""`roofline.traffic` of bench.py comes from committed counter passes
(profiles/traffic.json); VERDICT r3 item 6: a kernel change without a new PMC
pass must not keep a stale figure silently.  The record carries
`qmc_source_hash()` of the measured library; bench.py reports the figure only
when the library it runs has the same hash."""
import json
import os

import bench
from phd_qmclib_amd import _lib


def _record(tmp_path, sha):
    rec = {'head': 'abc', 'N64': {
        'vmc_step_kernel_bytes_per_chain_step': 1061.0,
        'vmc_step_kernel_valu_instr_per_chain_step': 800.0,
        'vmc_step_kernel_kernel_source_sha': sha,
        'dmc_evolve_kernel_bytes_per_walker_step': 2900.0,
        'dmc_evolve_kernel_kernel_source_sha': 'feedfeedfeedfeed'}}
    p = tmp_path / 'traffic.json'
    p.write_text(json.dumps(rec))
    return str(p)


def test_matching_hash_reports_the_figure(tmp_path):
    path = _record(tmp_path, 'aaaabbbbccccdddd')
    ent, note = bench.load_traffic(64, 'vmc_step_kernel', path=path,
                                   loaded='aaaabbbbccccdddd')
    assert note is None
    assert ent['bytes_per_chain_step'] == 1061.0
    assert ent['valu_instr_per_chain_step'] == 800.0


def test_stale_record_is_not_reported(tmp_path):
    path = _record(tmp_path, 'aaaabbbbccccdddd')
    ent, note = bench.load_traffic(64, 'vmc_step_kernel', path=path,
                                   loaded='0000111122223333')
    assert ent == {} and 'stale' in note
    # per kernel family: the DMC record of the same file has another hash
    ent, note = bench.load_traffic(64, 'dmc_evolve_kernel', path=path,
                                   loaded='aaaabbbbccccdddd')
    assert ent == {} and 'feedfeedfeedfeed' in note
    # a record without a hash (the round-3 file format) is stale by definition
    path = _record(tmp_path, None)
    ent, note = bench.load_traffic(64, 'vmc_step_kernel', path=path,
                                   loaded='aaaabbbbccccdddd')
    assert ent == {} and note
    ent, note = bench.load_traffic(128, 'vmc_step_kernel', path=path,
                                   loaded='aaaabbbbccccdddd')
    assert ent == {} and 'no traffic record' in note
    ent, note = bench.load_traffic(64, 'vmc_step_kernel',
                                   path=str(tmp_path / 'missing.json'),
                                   loaded='aaaabbbbccccdddd')
    assert ent == {} and note


def test_vmc_line_carries_null_traffic_and_a_note_when_stale(tmp_path,
                                                             monkeypatch):
    monkeypatch.setattr(bench, 'TRAFFIC_JSON', _record(tmp_path, 'not-this'))
    args = bench.parse_args(['--steps', '4'])
    m = dict(dt=1.0, kernel_ms=900.0, launches=4, launch_ms_avg_isolated=1.0,
             launch_ms_min=1.0, launch_ms_max=1.0, energy_per_particle=15.7,
             accept_rate=0.47)
    line = bench.vmc_line(args, m, 64, 1024, 1)
    assert line['roofline']['traffic'] is None
    assert 'stale' in line['roofline']['traffic_note']
    assert line['extra']['valu']['instr_per_chain_step'] is None
    # ... and the figure when the hash is the loaded library's
    monkeypatch.setattr(bench, 'TRAFFIC_JSON',
                        _record(tmp_path, _lib.source_hash()))
    line = bench.vmc_line(args, m, 64, 1024, 1)
    assert line['roofline']['traffic'] == 1061.0 * 1024
    assert 'same qmc_source_hash' in line['roofline']['traffic_note']


def test_library_hash_is_the_hash_of_the_sources():
    """The Makefile's recipe restated: sha256 over headers + .hip sources +
    flags, 16 hex digits -- the shipped library was built from the sources in
    the tree."""
    import hashlib
    import re
    import subprocess
    csrc = os.path.join(os.path.dirname(_lib.__file__), 'csrc')
    mk = open(os.path.join(csrc, 'Makefile')).read()
    hdrs = re.search(r'^HDRS\s*:=\s*(.*)$', mk, re.M).group(1).split()
    flags = re.search(r'^HIPFLAGS\s*\?=\s*(.*)$', mk, re.M).group(1)
    flags = flags.replace('$(ARCH)', 'gfx950')
    srcs = sorted(f for f in os.listdir(csrc) if f.endswith('.hip'))
    h = hashlib.sha256()
    for f in hdrs + srcs:
        h.update(open(os.path.join(csrc, f), 'rb').read())
    h.update((flags + '\n').encode())
    got = _lib.source_hash()
    if os.environ.get('QMCWALK_LIB'):
        return            # a variant build: other flags by construction
    assert got == h.hexdigest()[:16], \
        'libqmcwalk.so is older than its sources: run make -C phd_qmclib_amd/csrc'
    assert subprocess  # (quiet linters)
