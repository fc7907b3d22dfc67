"""Command-line front end: `python -m phd_qmclib_amd.mrbp_qmc.dmc_cli start
CONFIG.yml` (reference: mrbp_qmc/dmc_cli.py `start` command, console script
`mrbp-dmc`).  Runs every procedure of the configuration in order and writes
each result to the HDF5 location the configuration names."""
import argparse
import logging
import sys
from pathlib import Path

from .dmc_exec import CLIApp, config_loader

BANNER = '''
#####################################################################

    Diffusion Monte Carlo simulation for an interacting Bose gas
    within multi-rods with a contact interaction (MI355X engine).

#####################################################################
'''


def start(config_path, dry_run=False, verbose=False, silent=False):
    """Load the configuration, build the application and execute it."""
    config_path = Path(config_path).absolute()
    if not config_path.is_file():
        raise FileNotFoundError(config_path)
    if not silent:
        print(BANNER)
    logging.basicConfig(level=logging.DEBUG if verbose else logging.INFO,
                        format='%(asctime)s %(name)s %(message)s')
    app = CLIApp.from_config(config_loader.load(config_path))
    if dry_run:
        print(f'{len(app.app_spec)} procedure(s) validated; dry run, '
              f'nothing executed')
        return app, None
    results = app.exec()
    if not silent:
        print('Execution completed')
    return app, results


def main(argv=None):
    ap = argparse.ArgumentParser(prog='mrbp-dmc', description=__doc__)
    sub = ap.add_subparsers(dest='command', required=True)
    st = sub.add_parser('start', help='Start a Diffusion Monte Carlo simulation')
    st.add_argument('config_path')
    st.add_argument('-v', '--verbose', action='store_true')
    st.add_argument('-S', '--silent', action='store_true')
    st.add_argument('-y', '--assume-yes', action='store_true')
    st.add_argument('--dry-run', action='store_true')
    args = ap.parse_args(argv)
    start(args.config_path, args.dry_run, args.verbose, args.silent)
    return 0


if __name__ == '__main__':
    sys.exit(main())
