// qmcwalk.hip -- kernels and C-ABI of libqmcwalk.so (gfx950 / MI355X).
// Interface: include/qmcwalk.h.  Device building blocks: qmc_device.h.
//
// HBM layout (all fp64, "slot-major"): a walker slot owns one contiguous row
// of N positions and one of N drifts, pos[W][N] / drift[W][N]; the lane that
// owns particle i of walker w reads pos[w][i], so a lane group loads its row
// with one coalesced 8*G-byte access.  Per-walker scalars (energy, weight,
// slot energy, parent index) are plain [W] arrays.  DMC keeps two such
// population buffers (parents / children) that swap roles every time step.
#include "qmc_inst.h"
#include "qmc_kernels_misc.h"
#include "../../include/qmcwalk.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

// --------------------------------------------------------------- errors ---
static thread_local std::string g_err;

static int fail(const std::string &msg)
{
    g_err = msg;
    return 1;
}

#define HIP_TRY(expr)                                                        \
    do {                                                                     \
        hipError_t _e = (expr);                                              \
        if (_e != hipSuccess)                                                \
            return fail(std::string(#expr) + ": " + hipGetErrorString(_e));  \
    } while (0)

extern "C" const char *qmc_last_error(void) { return g_err.c_str(); }
extern "C" int qmc_abi_version(void) { return QMCWALK_ABI_VERSION; }
#ifndef QMC_SRC_SHA
#define QMC_SRC_SHA "unknown"
#endif
extern "C" const char *qmc_source_hash(void) { return QMC_SRC_SHA; }
extern "C" int qmc_device_count(int *count)
{
    HIP_TRY(hipGetDeviceCount(count));
    return 0;
}

// ------------------------------------------------------------- handles ----
struct qmc_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    DevModel dm;
    DevModel *dm_dev = nullptr;
    double *ob_table_dev = nullptr;     // one-body table rows (or null)
    double *trig_table_dev = nullptr;   // pair-angle row table (or null)
    unsigned long long *sec_prof_dev = nullptr;  // QMC_TIMING builds only
    unsigned long long *diag_dev = nullptr;      // DevModel::diag (QMC_NDIAG)
    qmc_model_params mp;
    int G = 64, P = 1;
    bool pad = false;
    bool fast = false;      // reduced-precision pair loop (float), off by default
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // kernel profile (qmc_engine_profile_begin/end): one event pair around
    // every launch of the dominant kernel (vmc_step / dmc_evolve)
    std::vector<hipEvent_t> prof_ev;
    size_t prof_used = 0;
    bool prof_on = false;
};

// Bracket a launch of the dominant kernel with an event pair while a profile
// is open (bench.py: roofline.achieved needs that kernel's own duration).
struct ProfScope {
    qmc_engine *e;
    bool on;
    explicit ProfScope(const qmc_engine *ce) : e(const_cast<qmc_engine *>(ce))
    {
        on = e->prof_on && e->prof_used + 2 <= e->prof_ev.size();
        if (on) (void)hipEventRecord(e->prof_ev[e->prof_used], e->stream);
    }
    ~ProfScope()
    {
        if (on) {
            (void)hipEventRecord(e->prof_ev[e->prof_used + 1], e->stream);
            e->prof_used += 2;
        }
    }
};


static int pick_shape(int n, int &G, int &P, bool &pad)
{
    if (n < 1) return 1;
    // tuning knob (tools/shape_bench.py): QMCWALK_SHAPE="G,P" forces a lane
    // group shape when it can hold the model (n <= G * P)
    if (const char *env = getenv("QMCWALK_SHAPE")) {
        int g = 0, p = 0;
        if (sscanf(env, "%d,%d", &g, &p) == 2 && n <= g * p) {
            static const int ok[][2] = { {16, 1}, {16, 2}, {32, 2}, {64, 1},
                                         {64, 2}, {64, 4}, {64, 8} };
            for (auto &s : ok)
                if (s[0] == g && s[1] == p) {
                    G = g; P = p; pad = (n != G * P);
                    return 0;
                }
        }
    }
    if (n <= 16) { G = 16; P = 1; }
    else if (n <= 32) { G = 16; P = 2; }
    else if (n <= 64) { G = 64; P = 1; }
    else if (n <= 128) { G = 64; P = 2; }
    else if (n <= 256) { G = 64; P = 4; }
    else if (n <= 512) { G = 64; P = 8; }
    else return 1;
    pad = (n != G * P);
    return 0;
}

// kernels are instantiated in inst_G_P.hip (one translation unit per shape)
#define QMC_EXTERN_TU(name) QMC_TU_##name(extern)
QMC_FOR_ALL_TUS(QMC_EXTERN_TU)
#undef QMC_EXTERN_TU

// ------------------------------------------------------------ dispatch ----
template <template <int, int, bool, bool> class L, typename... A>
static int dispatch_shape(const qmc_engine *e, A &&...args)
{
    const bool zc = e->dm.zclass != 0;
    // The masked variant is also the leaner one in registers (its per-pair
    // guards stop the compiler from keeping several pairs in flight: 110-160
    // VGPRs against 134-282 at P = 4, 8), and occupancy is what the large
    // shapes lack; each kernel family says from which P it wants it (and the
    // unmasked variants it never launches are not instantiated, qmc_inst.h).
#define QMC_CASE(g, p)                                                        \
    if (e->G == g && e->P == p) {                                             \
        constexpr bool always = L<g, p, true, false>::want_mask(p);           \
        if (always || e->pad) {                                               \
            if (zc) return L<g, p, true, true>::run(e, args...);              \
            return L<g, p, true, false>::run(e, args...);                     \
        }                                                                     \
        if constexpr (!always) {                                              \
            if (zc) return L<g, p, false, true>::run(e, args...);             \
            return L<g, p, false, false>::run(e, args...);                    \
        }                                                                     \
    }
    QMC_FOR_ALL_SHAPES(QMC_CASE)
#undef QMC_CASE
    return fail("unsupported boson_number");
}

template <int G, int P, bool ZC>
static size_t lds_bytes()
{
    return (size_t)(WalkBlock<G>::N / G) * GroupLds<G, P, ZC>::DOUBLES *
           sizeof(double);
}

// (the stepping kernels: StepLds, qmc_kernels.h)
template <int G, int P, bool PAD, bool ZC, bool DMC = false>
static size_t step_lds_bytes()
{
    return (size_t)(WalkBlock<G>::N / G) *
           StepLds<G, P, PAD, ZC, DMC>::DOUBLES * sizeof(double);
}

// Dynamic LDS above the default limit must be opted into per kernel.
template <typename K>
static void allow_lds(K kernel, size_t bytes)
{
    if (bytes > 48 * 1024)
        (void)hipFuncSetAttribute((const void *)kernel,
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)bytes);
}

template <int G>
static unsigned grid_for(long long nwalkers)
{
    const long long gpb = WalkBlock<G>::N / G;
    long long blocks = (nwalkers + gpb - 1) / gpb;
    // (one wavefront per workgroup: walker_of_block permutes runs of 128)
    if (QMC_XCD_MAP && gpb == 1) blocks = (blocks + 127) / 128 * 128;
    return (unsigned)blocks;
}

// The float pair loop exists for the one-wavefront-per-walker shapes with
// pairs classified from the sines (qmc_inst.h).
template <int G, bool ZC>
static constexpr bool has_fast() { return G == 64 && !ZC; }

template <int G, int P, bool PAD, bool ZC>
struct LaunchEval {
    static constexpr bool want_mask(int np) { return np >= 4; }
    static int run(const qmc_engine *e, const EvalArgs &a)
    {
        if (a.nconf <= 0) return 0;
        const size_t lds = lds_bytes<G, P, ZC>();
        if constexpr (has_fast<G, ZC>()) {
            if (e->fast) {
                allow_lds(evaluate_kernel<G, P, PAD, ZC, float>, lds);
                hipLaunchKernelGGL((evaluate_kernel<G, P, PAD, ZC, float>),
                                   dim3(grid_for<G>(a.nconf)), dim3(WalkBlock<G>::N),
                                   lds, e->stream, e->dm_dev, a);
                HIP_TRY(hipGetLastError());
                return 0;
            }
        }
        allow_lds(evaluate_kernel<G, P, PAD, ZC>, lds);
        hipLaunchKernelGGL((evaluate_kernel<G, P, PAD, ZC>),
                           dim3(grid_for<G>(a.nconf)), dim3(WalkBlock<G>::N),
                           lds, e->stream, e->dm_dev, a);
        HIP_TRY(hipGetLastError());
        return 0;
    }
};

template <int G, int P, bool PAD, bool ZC>
struct LaunchPrep {
    static constexpr bool want_mask(int np) { return np >= 4; }
    static int run(const qmc_engine *e, const PrepArgs &a)
    {
        if (a.nconf <= 0) return 0;
        const size_t lds = lds_bytes<G, P, ZC>();
        if constexpr (has_fast<G, ZC>()) {
            if (e->fast) {
                allow_lds(prepare_kernel<G, P, PAD, ZC, float>, lds);
                hipLaunchKernelGGL((prepare_kernel<G, P, PAD, ZC, float>),
                                   dim3(grid_for<G>(a.nconf)), dim3(WalkBlock<G>::N),
                                   lds, e->stream, e->dm_dev, a);
                HIP_TRY(hipGetLastError());
                return 0;
            }
        }
        allow_lds(prepare_kernel<G, P, PAD, ZC>, lds);
        hipLaunchKernelGGL((prepare_kernel<G, P, PAD, ZC>),
                           dim3(grid_for<G>(a.nconf)), dim3(WalkBlock<G>::N),
                           lds, e->stream, e->dm_dev, a);
        HIP_TRY(hipGetLastError());
        return 0;
    }
};

template <int G, int P, bool PAD, bool ZC>
struct LaunchVmc {
    static constexpr bool want_mask(int np) { return np >= 4; }
    static int run(const qmc_engine *e, const VmcArgs &a)
    {
        const size_t lds = step_lds_bytes<G, P, PAD, ZC>();
        const bool lean = !a.tape && !a.gaussian && !a.ser_wf && !a.ser_e &&
                          !a.ser_stat && !a.ser_pos;
        ProfScope prof(e);
        if constexpr (has_fast<G, ZC>()) {
            if (e->fast) {
                if (lean) {
                    allow_lds(vmc_step_kernel<G, P, PAD, ZC, true, float>, lds);
                    hipLaunchKernelGGL(
                        (vmc_step_kernel<G, P, PAD, ZC, true, float>),
                        dim3(grid_for<G>(a.W)), dim3(WalkBlock<G>::N), lds, e->stream,
                        e->dm_dev, a);
                } else {
                    allow_lds(vmc_step_kernel<G, P, PAD, ZC, false, float>,
                              lds);
                    hipLaunchKernelGGL(
                        (vmc_step_kernel<G, P, PAD, ZC, false, float>),
                        dim3(grid_for<G>(a.W)), dim3(WalkBlock<G>::N), lds, e->stream,
                        e->dm_dev, a);
                }
                HIP_TRY(hipGetLastError());
                return 0;
            }
        }
        if (lean) {
            allow_lds(vmc_step_kernel<G, P, PAD, ZC, true>, lds);
            hipLaunchKernelGGL((vmc_step_kernel<G, P, PAD, ZC, true>),
                               dim3(grid_for<G>(a.W)), dim3(WalkBlock<G>::N), lds,
                               e->stream, e->dm_dev, a);
        } else {
            allow_lds(vmc_step_kernel<G, P, PAD, ZC, false>, lds);
            hipLaunchKernelGGL((vmc_step_kernel<G, P, PAD, ZC, false>),
                               dim3(grid_for<G>(a.W)), dim3(WalkBlock<G>::N), lds,
                               e->stream, e->dm_dev, a);
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }
};

template <int G, int P, bool PAD, bool ZC>
struct LaunchEvolve {
    // (P = 8: the unmasked variant needs 253 registers and ran 1.7x slower,
    // profiles/r02_n512_tile_sweep.txt)
    static constexpr bool want_mask(int np) { return np >= 4; }
    static int run(const qmc_engine *e, const EvolveArgs &a)
    {
        const size_t lds = step_lds_bytes<G, P, PAD, ZC, true>();
        ProfScope prof(e);
        if constexpr (has_fast<G, ZC>()) {
            if (e->fast) {
                allow_lds(dmc_evolve_kernel<G, P, PAD, ZC, float>, lds);
                hipLaunchKernelGGL((dmc_evolve_kernel<G, P, PAD, ZC, float>),
                                   dim3(grid_for<G>(a.maxw)), dim3(WalkBlock<G>::N),
                                   lds, e->stream, e->dm_dev, a);
                HIP_TRY(hipGetLastError());
                return 0;
            }
        }
        allow_lds(dmc_evolve_kernel<G, P, PAD, ZC>, lds);
        hipLaunchKernelGGL((dmc_evolve_kernel<G, P, PAD, ZC>),
                           dim3(grid_for<G>(a.maxw)), dim3(WalkBlock<G>::N),
                           lds, e->stream, e->dm_dev, a);
        HIP_TRY(hipGetLastError());
        return 0;
    }
};

// -------------------------------------------------------------- engine ----
static void build_dev_model(const qmc_model_params &p, DevModel &d)
{
    memset(&d, 0, sizeof(d));
    d.n = (int)p.boson_number;
    d.is_free = p.is_free != 0;
    d.is_ideal = p.is_ideal != 0;
    d.defects_sep = (int)(p.defects_sep > 0 ? p.defects_sep : 1);
    d.L = p.supercell_size;
    d.half_L = 0.5 * d.L;
    d.rm = fabs(p.tbf_contact_cutoff);
    d.L_minus_rm = d.L - d.rm;
    d.two_over_L = 2.0 / d.L;
    d.k2_2pi = p.param_k2 * (2.0 / QMC_PI);
    d.k2 = p.param_k2;
    d.k2sq = d.k2 * d.k2;
    double phi = p.param_k2 * p.param_r_off;
    d.cphi = cos(phi);
    d.sphi = sin(phi);
    d.sec_cut = -1;
    d.am_cphi = fabs(p.param_am) * d.cphi;
    d.am_sphi = fabs(p.param_am) * d.sphi;
    d.m_k2cphi = -d.k2 * d.cphi;
    d.k2sphi = d.k2 * d.sphi;
    double th = p.param_k2 * d.L;
    d.cth = cos(th);
    d.sth = fabs(sin(th));
    d.sth_sign = sin(th) < 0.0 ? (int)0x80000000u : 0;
    d.sth_signed = sin(th);
    // leading all-short steps (qmc_device.h): theta = k2 D' - phi must identify
    // D' within a window of two box lengths
    d.sp_ok = (!d.is_ideal && th > 0.0 && th < QMC_PI - 1e-9 &&
               d.rm < d.half_L) ? 1 : 0;
    d.sp_xlo = -d.k2 * sin(d.k2 * d.rm - phi);
    d.sp_xhi = d.k2 * sin(phi);
    d.sp_cos = cos(d.k2 * d.rm);
    {
        // angles added to k2 z_own for the four short-range cases
        const double ang[4] = { -phi, phi, phi - th, th - phi };
        for (int v = 0; v < 4; ++v) {
            d.var_cos[v] = cos(ang[v]);
            d.var_sin[v] = sin(ang[v]);
        }
        d.m_k2 = -d.k2;
    }
    d.sin_rm = (d.rm >= d.half_L) ? 1.0 : sin(QMC_PI * d.rm / d.L);
#ifdef QMC_ABLATION     // timing experiments (tools/build_variant.sh): wrong physics
    if (const char *env = getenv("QMCWALK_ABL_SINRM")) d.sin_rm = atof(env);
#endif
    // sin(pi r / L) is flat near r = L/2: classify from positions there
    d.zclass = d.rm > 0.45 * d.L;
    double pi_L = QMC_PI / d.L;
    d.a_long = pi_L * p.param_beta;
    d.b_long = pi_L * pi_L * p.param_beta;
    d.a_long_sq = d.a_long * d.a_long;
    d.m_k2_over_a = d.a_long != 0.0 ? -d.k2 / d.a_long : 0.0;
    d.m_a_over_k2 = d.k2 != 0.0 ? -d.a_long / d.k2 : 0.0;
    d.beta = p.param_beta;
    d.inv_beta = p.param_beta != 0.0 ? 1.0 / p.param_beta : 0.0;
    d.one_minus_beta = 1.0 - p.param_beta;
    d.log_am = log(fabs(p.param_am));
    d.z_a = 1.0 / (1.0 + p.lattice_ratio);
    d.z_b = p.lattice_ratio / (1.0 + p.lattice_ratio);
    d.k1 = p.param_k1;
    d.k1_2pi = p.param_k1 * (2.0 / QMC_PI);
    d.kp1 = p.param_kp1;
    d.e0 = p.param_e0;
    d.v0 = p.lattice_depth;
    d.v0d = p.defect_magnitude;
    d.uniform_barrier = (d.defects_sep == 1 || d.v0d == d.v0) ? 1 : 0;
    d.v_barrier = (d.defects_sep == 1) ? d.v0d : d.v0;
    d.k1_half = 0.5 * p.param_k1;
    d.v0_minus_e0 = p.lattice_depth - p.param_e0;
    if (!d.is_free) {
        double sh = sinh(0.5 * sqrt(d.v0 - d.e0) * d.z_b);
        d.cf = sqrt(1 + d.v0 / d.e0 * sh * sh);
    } else {
        d.cf = 1.0;
    }
}

// ---- one-body table (qmc_device.h one_body_tab) ---------------------------
// Closed forms of mrbp_qmc/model.py:404-464 in long double.
static void ob_exact(const DevModel &d, long double zc, bool barrier,
                     long double &ldz, long double &logf)
{
    if (barrier) {
        const long double x = (long double)d.kp1 *
                              (zc - 1.0L + 0.5L * (long double)d.z_b);
        ldz = (long double)d.kp1 * tanhl(x);
        logf = logl(coshl(x));
    } else {
        const long double x = (long double)d.k1 *
                              (zc - 0.5L * (long double)d.z_a);
        ldz = -(long double)d.k1 * tanl(x);
        logf = logl((long double)d.cf * cosl(x));
    }
}

// Degree-DEG interpolant of f on [a, a + h] at Chebyshev nodes, as monomial
// coefficients in t = (x - a) / h (long double elimination, partial pivoting).
template <int DEG, typename F>
static void ob_fit(F f, long double a, long double h, double *coef)
{
    constexpr int K = DEG + 1;
    long double A[K][K + 1];
    for (int j = 0; j < K; ++j) {
        const long double t = 0.5L * (1.0L + cosl(3.14159265358979323846264338L *
                                                  (2 * j + 1) / (2.0L * K)));
        long double pw = 1.0L;
        for (int k = 0; k < K; ++k) { A[j][k] = pw; pw *= t; }
        A[j][K] = f(a + h * t);
    }
    for (int c = 0; c < K; ++c) {
        int piv = c;
        for (int r = c + 1; r < K; ++r)
            if (fabsl(A[r][c]) > fabsl(A[piv][c])) piv = r;
        for (int k = 0; k <= K; ++k) std::swap(A[c][k], A[piv][k]);
        for (int r = c + 1; r < K; ++r) {
            const long double g = A[r][c] / A[c][c];
            for (int k = c; k <= K; ++k) A[r][k] -= g * A[c][k];
        }
    }
    for (int r = K - 1; r >= 0; --r) {
        long double v = A[r][K];
        for (int k = r + 1; k < K; ++k) v -= A[r][k] * (long double)coef[k];
        coef[r] = (double)(v / A[r][r]);
    }
}

static double ob_horner(const double *c, int deg, double t)
{
    double p = c[deg];
    for (int k = deg - 1; k >= 0; --k) p = fma(p, t, c[k]);
    return p;
}

// Rows of one lattice region at `m` intervals; -> worst error of the double
// Horner evaluation against the closed forms, relative to max(1, |f|).
static double ob_build_region(const DevModel &d, bool barrier, int m,
                              double *rows)
{
    const long double lo = barrier ? (long double)d.z_a : 0.0L;
    const long double len = barrier ? 1.0L - (long double)d.z_a
                                    : (long double)d.z_a;
    const long double h = len / m;
    double worst = 0.0;
    for (int i = 0; i < m; ++i) {
        const long double a = lo + h * i;
        double *r = rows + (size_t)i * OB_ROW;
        ob_fit<OB_DEG>([&](long double x) {
            long double l, g; ob_exact(d, x, barrier, l, g); return l; },
            a, h, r);
        ob_fit<OB_DEG>([&](long double x) {
            long double l, g; ob_exact(d, x, barrier, l, g); return g; },
            a, h, r + 8);
        for (int q = 0; q <= 12; ++q) {
            const double t = q / 12.0;
            long double l, g;
            ob_exact(d, a + h * (long double)t, barrier, l, g);
            const double e1 = fabs(ob_horner(r, OB_DEG, t) - (double)l) /
                              fmax(1.0, fabs((double)l));
            const double e2 = fabs(ob_horner(r + 8, OB_DEG, t) - (double)g) /
                              fmax(1.0, fabs((double)g));
            worst = fmax(worst, fmax(e1, e2));
        }
    }
    return worst;
}

static double g_ob_last_err[2] = { 0.0, 0.0 };   // diagnostics (info call)
static constexpr double OB_TOL = 2e-15;
static constexpr int OB_MAX_ROWS = 1024;        // per region (128 KB)

// -> host table (empty when the model does not reach OB_TOL: the kernels then
// evaluate the closed forms directly) and the interval counts.
static void build_ob_table(const DevModel &d, std::vector<double> &tab,
                           int &m1, int &m2)
{
    tab.clear();
    m1 = m2 = 0;
    if (d.is_free) return;
    if (const char *env = getenv("QMCWALK_OB_TABLE"))
        if (env[0] == '0') return;              // tuning / A-B knob
    if (!(d.z_a > 0.0 && d.z_a < 1.0) || !std::isfinite(d.k1) ||
        !std::isfinite(d.kp1) || !std::isfinite(d.cf))
        return;
    std::vector<double> reg[2];
    int m[2] = { 0, 0 };
    for (int b = 0; b < 2; ++b) {
        for (int cand = 16; cand <= OB_MAX_ROWS; cand *= 2) {
            reg[b].assign((size_t)cand * OB_ROW, 0.0);
            const double err = ob_build_region(d, b == 1, cand, reg[b].data());
            g_ob_last_err[b] = err;
            if (std::isfinite(err) && err <= OB_TOL) { m[b] = cand; break; }
        }
        if (!m[b]) return;
    }
    // the row map of the kernel is the larger of two lines: the barrier's
    // intervals must not be wider than the well's
    const int need2 = (int)std::ceil((double)m[0] * (1.0 - d.z_a) / d.z_a);
    if (need2 > m[1]) {
        if (need2 > 8 * OB_MAX_ROWS) return;
        m[1] = need2;
        reg[1].assign((size_t)m[1] * OB_ROW, 0.0);
        const double err = ob_build_region(d, true, m[1], reg[1].data());
        g_ob_last_err[1] = err;
        if (!(std::isfinite(err) && err <= OB_TOL)) return;
    }
    m1 = m[0]; m2 = m[1];
    tab = reg[0];
    tab.insert(tab.end(), reg[1].begin(), reg[1].end());
    // closing row: z_cell -> 1 rounds into it at t = 0, where the periodic
    // factor continues with the first well interval
    tab.insert(tab.end(), reg[0].begin(), reg[0].begin() + OB_ROW);
}

// Row table of the pair-table angles (qmc_device.h trig_tab): the smallest
// power-of-two row count that keeps both angle offsets inside QMC_TRIG_DMAX,
// nothing if that takes more than QMC_TRIG_MAX_ROWS rows (128 KB).
#define QMC_TRIG_DMAX 4.0e-3
#define QMC_TRIG_MAX_ROWS 4096
#ifndef QMC_TRIG_TABLE
#define QMC_TRIG_TABLE 1       // 0: build variant without it (A/B runs)
#endif
static void build_trig_table(DevModel &d, std::vector<double> &tab)
{
    tab.clear();
    d.trig_table = nullptr;
    d.tg_rows = 0;
    if (d.is_ideal || !(d.L > 0.0) || !QMC_TRIG_TABLE) return;
    const double span = fmax(QMC_PI, fabs(d.k2) * d.L);
    int rows = 256;
    while (rows <= QMC_TRIG_MAX_ROWS && span / (2.0 * rows) > QMC_TRIG_DMAX)
        rows *= 2;
    if (rows > QMC_TRIG_MAX_ROWS) return;
    const double h = d.L / rows;
    tab.resize((size_t)rows * 4);
    const long double pil = 3.141592653589793238462643383279502884L;
    for (int r = 0; r < rows; ++r) {
        const long double zr = ((long double)r + 0.5L) * (long double)h;
        const long double a1 = pil * zr / (long double)d.L;
        const long double a2 = (long double)d.k2 * zr;
        tab[4 * r + 0] = (double)sinl(a1);
        tab[4 * r + 1] = (double)cosl(a1);
        tab[4 * r + 2] = (double)sinl(a2);
        tab[4 * r + 3] = (double)cosl(a2);
    }
    d.tg_rows = rows;
    d.tg_h = h;
    d.tg_inv_h = (double)rows / d.L;
    d.tg_a1 = QMC_PI / d.L;
    d.tg_b1 = -d.tg_a1 * 0.5 * h;
    d.tg_a2 = d.k2;
    d.tg_b2 = -d.k2 * 0.5 * h;
}

// Diagnostic, no GPU needed: the table a model would get.
extern "C" int qmc_model_one_body_table_info(const qmc_model_params *model,
                                             int32_t *rows_well,
                                             int32_t *rows_barrier,
                                             double *max_err)
{
    if (!model) return fail("qmc_model_one_body_table_info: null argument");
    DevModel d;
    build_dev_model(*model, d);
    std::vector<double> tab;
    int m1 = 0, m2 = 0;
    g_ob_last_err[0] = g_ob_last_err[1] = 0.0;
    build_ob_table(d, tab, m1, m2);
    if (rows_well) *rows_well = m1;
    if (rows_barrier) *rows_barrier = m2;
    if (max_err) *max_err = fmax(g_ob_last_err[0], g_ob_last_err[1]);
    return 0;
}

static int engine_create_impl(const qmc_model_params *model, int device,
                              void *stream, bool caller_stream,
                              qmc_engine **out)
{
    if (!model || !out) return fail("qmc_engine_create: null argument");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail("qmc_engine_create: no HIP device");
    if (device < 0 || device >= ndev)
        return fail("qmc_engine_create: bad device index");
    qmc_engine *e = new qmc_engine();
    e->device = device;
    e->mp = *model;
    if (pick_shape((int)model->boson_number, e->G, e->P, e->pad)) {
        delete e;
        return fail("qmc_engine_create: boson_number must be in [1, 512]");
    }
    build_dev_model(*model, e->dm);
    // lanes of a group the rotation runs over: all of them when the model
    // fills the shape, else the smallest even number that holds it
    e->dm.ge = e->pad ? 2 * ((e->dm.n + 2 * e->P - 1) / (2 * e->P)) : e->G;
    // domain of the short-range kernel (qmc_device.h pair_core): the matching
    // conditions of mrbp_qmc/model.py:340-392 give phi = k2 r_off in (0, pi/2)
    // and k2 rm in (0, pi/2) for every repulsive model
    if (!e->dm.is_ideal &&
        (e->dm.sphi < 0.0 || e->dm.cphi < 0.0 ||
         e->dm.k2 * e->dm.rm >= 0.5 * QMC_PI || e->dm.k2 <= 0.0)) {
        delete e;
        return fail("qmc_engine_create: two-body parameters outside the "
                    "model's domain (need 0 < k2 rm < pi/2, 0 <= k2 r_off <= pi/2)");
    }
    HIP_TRY(hipSetDevice(device));
    if (caller_stream) {
        // the caller's stream as it is; NULL is the legacy default stream
        e->stream = (hipStream_t)stream;
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
        e->own_stream = true;
    }
    HIP_TRY(hipEventCreate(&e->ev0));
    HIP_TRY(hipEventCreate(&e->ev1));
    {
        std::vector<double> tab;
        int m1 = 0, m2 = 0;
        build_ob_table(e->dm, tab, m1, m2);
        if (!tab.empty()) {
            HIP_TRY(hipMalloc((void **)&e->ob_table_dev,
                              tab.size() * sizeof(double)));
            HIP_TRY(hipMemcpy(e->ob_table_dev, tab.data(),
                              tab.size() * sizeof(double),
                              hipMemcpyHostToDevice));
            DevModel &d = e->dm;
            d.ob_table = e->ob_table_dev;
            d.ob_m1 = m1; d.ob_m2 = m2;
            d.ob_invh1 = (double)m1 / d.z_a;
            d.ob_invh2 = (double)m2 / (1.0 - d.z_a);
            d.ob_shift2 = (double)m1 / d.ob_invh2 - d.z_a;
        }
    }
    {
        std::vector<double> tab;
        build_trig_table(e->dm, tab);
        if (!tab.empty()) {
            HIP_TRY(hipMalloc((void **)&e->trig_table_dev,
                              tab.size() * sizeof(double)));
            HIP_TRY(hipMemcpy(e->trig_table_dev, tab.data(),
                              tab.size() * sizeof(double),
                              hipMemcpyHostToDevice));
            e->dm.trig_table = e->trig_table_dev;
        }
    }
#if defined(QMC_TIMING)
    HIP_TRY(hipMalloc((void **)&e->sec_prof_dev, (size_t)QMC_SEC_COPIES * 2 *
                      QMC_NSEC * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(e->sec_prof_dev, 0, (size_t)QMC_SEC_COPIES * 2 *
                      QMC_NSEC * sizeof(unsigned long long)));
    e->dm.sec_prof = e->sec_prof_dev;
#endif
    HIP_TRY(hipMalloc((void **)&e->diag_dev,
                      QMC_NDIAG * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(e->diag_dev, 0, QMC_NDIAG * sizeof(unsigned long long)));
    e->dm.diag = e->diag_dev;
    HIP_TRY(hipMalloc((void **)&e->dm_dev, sizeof(DevModel)));
    HIP_TRY(hipMemcpy(e->dm_dev, &e->dm, sizeof(DevModel),
                      hipMemcpyHostToDevice));
    *out = e;
    return 0;
}

// Diagnostic, no GPU needed: the pair-angle row table a model would get and
// the worst deviation of trig_tab's arithmetic (restated here on the host,
// operation for operation) from long-double sin / cos over [0, L).
static void trig_small_host(double d, double &s, double &c)
{
    const double d2 = d * d;
    s = fma(d * d2, fma(d2, 1.0 / 120.0, -1.0 / 6.0), d);
    c = fma(d2, fma(d2, 1.0 / 24.0, -0.5), 1.0);
}

extern "C" int qmc_model_trig_table_info(const qmc_model_params *model,
                                         int32_t *rows, double *max_err)
{
    if (!model) return fail("qmc_model_trig_table_info: null argument");
    DevModel d;
    build_dev_model(*model, d);
    std::vector<double> tab;
    build_trig_table(d, tab);
    if (rows) *rows = d.tg_rows;
    double worst = 0.0;
    if (d.tg_rows) {
        const long double pil = 3.141592653589793238462643383279502884L;
        const int samples = 7;            // per row, edges included
        for (int r = 0; r < d.tg_rows; ++r) {
            for (int j = 0; j <= samples; ++j) {
                double z = ((double)r + (double)j / samples) * d.tg_h;
                if (j == samples) z = nextafter(z, 0.0);
                const int rr = (int)(z * d.tg_inv_h);
                if (rr < 0 || rr >= d.tg_rows) continue;   // device falls back
                const double dz = fma(-(double)rr, d.tg_h, z);
                const double *row = &tab[4 * (size_t)rr];
                double sd, cd, got[4];
                trig_small_host(fma(dz, d.tg_a1, d.tg_b1), sd, cd);
                got[0] = fma(row[0], cd, row[1] * sd);
                got[1] = fma(row[1], cd, -(row[0] * sd));
                trig_small_host(fma(dz, d.tg_a2, d.tg_b2), sd, cd);
                got[2] = fma(row[2], cd, row[3] * sd);
                got[3] = fma(row[3], cd, -(row[2] * sd));
                const long double a1 = pil * (long double)z / (long double)d.L;
                const long double a2 = (long double)d.k2 * (long double)z;
                const long double ref[4] = { sinl(a1), cosl(a1), sinl(a2),
                                             cosl(a2) };
                for (int k = 0; k < 4; ++k)
                    worst = fmax(worst, (double)fabsl((long double)got[k] - ref[k]));
            }
        }
    }
    if (max_err) *max_err = worst;
    return 0;
}

// Diagnostic, no GPU needed: log_pos (qmc_math.h) restated on the host,
// operation for operation, over 10^5 arguments between 1e-300 and 1e300 and
// a dense sweep around 1; worst |got - log x| / (1 + |log x|) against long
// double.
static const double g_log_tab_host[2 * QMC_LOG_ROWS] = { QMC_LOG_TAB_VALUES };

static double log_pos_host(double x)
{
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    int k;
    const double m = frexp(x, &k);                         // [1/2, 1)
    const int r = (int)fma(m, (double)(2 * QMC_LOG_ROWS), -(double)QMC_LOG_ROWS);
    const double inv_c = g_log_tab_host[2 * r], L = g_log_tab_host[2 * r + 1];
    const double d = fma(m, inv_c, -1.0);
    double q = fma(d, 0.2, -0.25);
    q = fma(q, d, 1.0 / 3.0);
    q = fma(q, d, -0.5);
    const double lm = fma(d * d, q, d) + L;
    const double kd = (double)k;
    return fma(kd, LN2_HI, fma(kd, LN2_LO, lm));
}

extern "C" int qmc_log_table_info(int32_t *rows, double *max_err)
{
    if (rows) *rows = QMC_LOG_ROWS;
    double worst = 0.0;
    uint64_t state = 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < 200000; ++i) {
        state = state * 6364136223846793005ull + 1442695040888963407ull;
        const double u = (double)(state >> 11) * (1.0 / 9007199254740992.0);
        double x;
        if (i < 100000) x = pow(10.0, -300.0 + 600.0 * u);          // any size
        else if (i < 150000) x = 1.0 + (u - 0.5) * 1e-3 * (i % 1000); // near 1
        else x = u + 1e-17;                                          // (0, 1)
        const long double ref = logl((long double)x);
        const double err = (double)(fabsl((long double)log_pos_host(x) - ref) /
                                    (1.0L + fabsl(ref)));
        if (err > worst) worst = err;
    }
    // row edges
    for (int r = 0; r <= QMC_LOG_ROWS; ++r) {
        for (int s = -1; s <= 1; ++s) {
            double x = QMC_LOG_LO + (double)r / QMC_LOG_INVW;
            if (s < 0) x = nextafter(x, 0.0);
            if (s > 0) x = nextafter(x, 2.0);
            const long double ref = logl((long double)x);
            const double err = (double)(fabsl((long double)log_pos_host(x) - ref) /
                                        (1.0L + fabsl(ref)));
            if (err > worst) worst = err;
        }
    }
    if (max_err) *max_err = worst;
    return 0;
}

extern "C" int qmc_engine_create(const qmc_model_params *model, int device,
                                 void *stream, qmc_engine **out)
{
    return engine_create_impl(model, device, stream, stream != nullptr, out);
}

extern "C" int qmc_engine_create_on_stream(const qmc_model_params *model,
                                           int device, void *stream,
                                           qmc_engine **out)
{
    return engine_create_impl(model, device, stream, true, out);
}

extern "C" int qmc_engine_set_fast_math(qmc_engine *e, int on, int *in_effect)
{
    if (!e) return fail("qmc_engine_set_fast_math: null engine");
    // available for the one-wavefront-per-walker shapes (boson_number > 32)
    // unless pairs are classified from positions (cutoff close to L/2);
    // elsewhere the request is a no-op: the knob grants a permission
    const bool can = e->G == 64 && !e->dm.zclass && !e->dm.is_ideal;
    e->fast = on && can;
    if (in_effect) *in_effect = e->fast ? 1 : 0;
    return 0;
}

extern "C" int qmc_engine_stream(qmc_engine *e, void **stream, int *owned)
{
    if (!e || !stream) return fail("qmc_engine_stream: null argument");
    *stream = (void *)e->stream;
    if (owned) *owned = e->own_stream ? 1 : 0;
    return 0;
}

extern "C" int qmc_engine_profile_begin(qmc_engine *e, int64_t max_launches)
{
    if (!e) return fail("qmc_engine_profile_begin: null engine");
    if (max_launches <= 0 || max_launches > (1 << 20))
        return fail("qmc_engine_profile_begin: max_launches out of range");
    HIP_TRY(hipSetDevice(e->device));
    while (e->prof_ev.size() < (size_t)max_launches * 2) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreate(&ev));
        e->prof_ev.push_back(ev);
    }
    e->prof_used = 0;
    e->prof_on = true;
    return 0;
}

extern "C" int qmc_engine_profile_end(qmc_engine *e, int64_t *launches,
                                      double *total_ms, double *min_ms,
                                      double *max_ms)
{
    if (!e) return fail("qmc_engine_profile_end: null engine");
    HIP_TRY(hipSetDevice(e->device));
    e->prof_on = false;
    double tot = 0.0, mn = 0.0, mx = 0.0;
    const size_t n = e->prof_used / 2;
    if (n) HIP_TRY(hipEventSynchronize(e->prof_ev[e->prof_used - 1]));
    for (size_t i = 0; i < n; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e->prof_ev[2 * i],
                                    e->prof_ev[2 * i + 1]));
        tot += ms;
        if (i == 0 || ms < mn) mn = ms;
        if (i == 0 || ms > mx) mx = ms;
    }
    e->prof_used = 0;
    if (launches) *launches = (int64_t)n;
    if (total_ms) *total_ms = tot;
    if (min_ms) *min_ms = mn;
    if (max_ms) *max_ms = mx;
    return 0;
}

// Diagnostic libraries built with -DQMC_TIMING (tools/section_times.py): cycles
// of wavefront lifetime and visits per kernel section since the last reset,
// summed over every walker kernel launched on this engine.  Section i of the
// first pass of a kernel is qmc_section_name(i); i + nsec / 2 is the same
// section inside the energy pass of the VMC step.  The shipped library has no
// stamps in its kernels and returns an error here.
extern "C" int qmc_engine_section_profile(qmc_engine *e, uint64_t *cycles,
                                          uint64_t *visits, int32_t nsec,
                                          int32_t reset)
{
    if (!e) return fail("qmc_engine_section_profile: null engine");
#if defined(QMC_TIMING)
    if (nsec != QMC_NSEC)
        return fail("qmc_engine_section_profile: nsec must be 32");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    std::vector<unsigned long long> host((size_t)QMC_SEC_COPIES * 2 * QMC_NSEC);
    const size_t bytes = host.size() * sizeof(unsigned long long);
    HIP_TRY(hipMemcpy(host.data(), e->sec_prof_dev, bytes,
                      hipMemcpyDeviceToHost));
    for (int i = 0; i < QMC_NSEC; ++i) {
        unsigned long long c = 0, v = 0;
        for (int k = 0; k < QMC_SEC_COPIES; ++k) {
            c += host[(size_t)k * 2 * QMC_NSEC + i];
            v += host[(size_t)k * 2 * QMC_NSEC + QMC_NSEC + i];
        }
        if (cycles) cycles[i] = c;
        if (visits) visits[i] = v;
    }
    if (reset) HIP_TRY(hipMemset(e->sec_prof_dev, 0, bytes));
    return 0;
#else
    (void)cycles; (void)visits; (void)nsec; (void)reset;
    return fail("qmc_engine_section_profile: this library was built without "
                "-DQMC_TIMING (see tools/section_times.py)");
#endif
}

// Diagnostic libraries built with -DQMC_CUTS (tools/section_counts.py): from
// now on every wavefront of the walker kernels ends when it reaches section
// mark `section` (-1: never).  The results of such launches are meaningless;
// hardware counters of runs cut at successive marks give the instructions each
// section executes.  The shipped library has no cut tests in its kernels.
extern "C" int qmc_engine_section_cut(qmc_engine *e, int32_t section)
{
    if (!e) return fail("qmc_engine_section_cut: null engine");
#if defined(QMC_CUTS)
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->dm.sec_cut = section;
    HIP_TRY(hipMemcpy(e->dm_dev, &e->dm, sizeof(DevModel),
                      hipMemcpyHostToDevice));
    return 0;
#else
    (void)section;
    return fail("qmc_engine_section_cut: this library was built without "
                "-DQMC_CUTS (see tools/section_counts.py)");
#endif
}

// Diagnostic counters the kernels keep (DevModel::diag); synchronises.
extern "C" int qmc_engine_diag_counters(qmc_engine *e, uint64_t *out,
                                        int32_t n, int32_t reset)
{
    if (!e) return fail("qmc_engine_diag_counters: null engine");
    if (n < 0 || n > QMC_NDIAG)
        return fail("qmc_engine_diag_counters: at most 4 counters");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (out && n > 0)
        HIP_TRY(hipMemcpy(out, e->diag_dev, (size_t)n * sizeof(uint64_t),
                          hipMemcpyDeviceToHost));
    if (reset)
        HIP_TRY(hipMemset(e->diag_dev, 0,
                          QMC_NDIAG * sizeof(unsigned long long)));
    return 0;
}

extern "C" const char *qmc_section_name(int32_t i)
{
    if (i < 0 || i >= QMC_NSEC) return "";
    return QMC_SEC_NAMES[i % (QMC_NSEC / 2)];
}

extern "C" void qmc_engine_destroy(qmc_engine *e)
{
    if (!e) return;
    hipSetDevice(e->device);
    for (hipEvent_t ev : e->prof_ev) hipEventDestroy(ev);
    if (e->ev0) hipEventDestroy(e->ev0);
    if (e->ev1) hipEventDestroy(e->ev1);
    if (e->dm_dev) hipFree(e->dm_dev);
    if (e->ob_table_dev) hipFree(e->ob_table_dev);
    if (e->trig_table_dev) hipFree(e->trig_table_dev);
    if (e->sec_prof_dev) hipFree(e->sec_prof_dev);
    if (e->diag_dev) hipFree(e->diag_dev);
    if (e->own_stream && e->stream) hipStreamDestroy(e->stream);
    delete e;
}

extern "C" int qmc_engine_sync(qmc_engine *e)
{
    if (!e) return fail("null engine");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return 0;
}

extern "C" int qmc_engine_timer_start(qmc_engine *e)
{
    if (!e) return fail("null engine");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipEventRecord(e->ev0, e->stream));
    return 0;
}

extern "C" int qmc_engine_timer_stop(qmc_engine *e, float *ms)
{
    if (!e || !ms) return fail("null argument");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipEventRecord(e->ev1, e->stream));
    HIP_TRY(hipEventSynchronize(e->ev1));
    HIP_TRY(hipEventElapsedTime(ms, e->ev0, e->ev1));
    return 0;
}

template <typename T>
static int dev_alloc(T **p, size_t count)
{
    *p = nullptr;
    if (count == 0) count = 1;
    HIP_TRY(hipMalloc((void **)p, count * sizeof(T)));
    return 0;
}

extern "C" int qmc_evaluate_dev(qmc_engine *e, int64_t nconf,
                                const double *pos, double *wf, double *energy,
                                double *ith, double *drift)
{
    if (!e || !pos) return fail("qmc_evaluate_dev: null argument");
    HIP_TRY(hipSetDevice(e->device));
    EvalArgs a{ pos, wf, energy, ith, drift, (long long)nconf };
    return dispatch_shape<LaunchEval>(e, a);
}

// Plain device buffers (configuration sets kept resident across
// qmc_evaluate_dev calls by callers without a device-memory library).
extern "C" int qmc_buffer_alloc(int device, size_t bytes, void **out)
{
    if (!out || !bytes) return fail("qmc_buffer_alloc: null argument");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMalloc(out, bytes));
    return 0;
}

extern "C" int qmc_buffer_free(void *buf)
{
    if (buf) HIP_TRY(hipFree(buf));
    return 0;
}

extern "C" int qmc_buffer_upload(void *dst_dev, const void *src_host,
                                 size_t bytes)
{
    if (!dst_dev || !src_host) return fail("qmc_buffer_upload: null argument");
    HIP_TRY(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int qmc_buffer_download(void *dst_host, const void *src_dev,
                                   size_t bytes)
{
    if (!dst_host || !src_dev)
        return fail("qmc_buffer_download: null argument");
    HIP_TRY(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int qmc_evaluate(qmc_engine *e, int64_t nconf, const double *pos,
                            double *wf, double *energy, double *ith,
                            double *drift)
{
    if (!e || !pos) return fail("qmc_evaluate: null argument");
    if (nconf <= 0) return 0;
    HIP_TRY(hipSetDevice(e->device));
    const size_t n = (size_t)e->dm.n, W = (size_t)nconf;
    double *dpos, *dwf, *den, *dith, *ddr;
    if (dev_alloc(&dpos, W * n) || dev_alloc(&dwf, W) || dev_alloc(&den, W) ||
        dev_alloc(&dith, W * n) || dev_alloc(&ddr, W * n))
        return 1;
    HIP_TRY(hipMemcpyAsync(dpos, pos, W * n * sizeof(double),
                           hipMemcpyHostToDevice, e->stream));
    int rc = qmc_evaluate_dev(e, nconf, dpos, dwf, den, dith, ddr);
    if (!rc) {
        if (wf) HIP_TRY(hipMemcpyAsync(wf, dwf, W * sizeof(double),
                                       hipMemcpyDeviceToHost, e->stream));
        if (energy) HIP_TRY(hipMemcpyAsync(energy, den, W * sizeof(double),
                                           hipMemcpyDeviceToHost, e->stream));
        if (ith) HIP_TRY(hipMemcpyAsync(ith, dith, W * n * sizeof(double),
                                        hipMemcpyDeviceToHost, e->stream));
        if (drift) HIP_TRY(hipMemcpyAsync(drift, ddr, W * n * sizeof(double),
                                          hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    hipFree(dpos); hipFree(dwf); hipFree(den); hipFree(dith); hipFree(ddr);
    return rc;
}

// Sort every configuration by position; label[w][lane] = original particle
// index held by the lane (see resort_step in qmc_device.h).
static void sort_rows(const double *pos, size_t W, size_t n,
                      std::vector<double> &sorted,
                      std::vector<unsigned short> &label,
                      const double *extra = nullptr,
                      std::vector<double> *extra_sorted = nullptr,
                      size_t pos_stride = 0, size_t extra_stride = 0)
{
    if (!pos_stride) pos_stride = n;
    if (!extra_stride) extra_stride = n;
    sorted.resize(W * n);
    label.resize(W * n);
    if (extra_sorted) extra_sorted->resize(W * n);
    std::vector<unsigned short> idx(n);
    for (size_t w = 0; w < W; ++w) {
        const double *row = pos + w * pos_stride;
        for (size_t i = 0; i < n; ++i) idx[i] = (unsigned short)i;
        std::stable_sort(idx.begin(), idx.end(),
                         [row](unsigned short a, unsigned short b) {
                             return row[a] < row[b];
                         });
        for (size_t i = 0; i < n; ++i) {
            sorted[w * n + i] = row[idx[i]];
            label[w * n + i] = idx[i];
            if (extra_sorted)
                (*extra_sorted)[w * n + i] = extra[w * extra_stride + idx[i]];
        }
    }
}

// ----------------------------------------------------------------- VMC ----
struct qmc_vmc {
    qmc_engine *eng = nullptr;
    qmc_vmc_params p;
    long long W = 0;
    double *pos = nullptr, *wf = nullptr, *ecarry = nullptr;
    unsigned short *label = nullptr;
    double *sum_e = nullptr, *sum_e2 = nullptr;
    long long *n_acc = nullptr;
    double *tape = nullptr;
    long long tape_steps = 0, tape_used = 0;
    unsigned int step = 0;
    int yield_initial = 1;
    double *ssf_partial = nullptr, *ssf_out = nullptr;   // qmc_vmc_ssf scratch
    int ssf_cap = 0;
};

extern "C" int qmc_vmc_create(qmc_engine *e, const qmc_vmc_params *p,
                              qmc_vmc **out)
{
    if (!e || !p || !out) return fail("qmc_vmc_create: null argument");
    if (p->num_chains <= 0) return fail("qmc_vmc_create: num_chains <= 0");
    HIP_TRY(hipSetDevice(e->device));
    qmc_vmc *v = new qmc_vmc();
    v->eng = e;
    v->p = *p;
    v->W = p->num_chains;
    const size_t W = (size_t)v->W, n = (size_t)e->dm.n;
    if (dev_alloc(&v->pos, W * n) || dev_alloc(&v->label, W * n) ||
        dev_alloc(&v->wf, W) || dev_alloc(&v->ecarry, W) || dev_alloc(&v->sum_e, W) ||
        dev_alloc(&v->sum_e2, W) || dev_alloc(&v->n_acc, W)) {
        delete v;
        return 1;
    }
    *out = v;
    return 0;
}

extern "C" void qmc_vmc_destroy(qmc_vmc *v)
{
    if (!v) return;
    hipSetDevice(v->eng->device);
    hipFree(v->pos); hipFree(v->label); hipFree(v->wf); hipFree(v->ecarry);
    if (v->ssf_partial) hipFree(v->ssf_partial);
    if (v->ssf_out) hipFree(v->ssf_out);
    hipFree(v->sum_e); hipFree(v->sum_e2); hipFree(v->n_acc);
    if (v->tape) hipFree(v->tape);
    delete v;
}

extern "C" int qmc_vmc_set_state(qmc_vmc *v, const double *pos)
{
    if (!v || !pos) return fail("qmc_vmc_set_state: null argument");
    qmc_engine *e = v->eng;
    HIP_TRY(hipSetDevice(e->device));
    const size_t W = (size_t)v->W, n = (size_t)e->dm.n;
    std::vector<double> sorted;
    std::vector<unsigned short> label;
    sort_rows(pos, W, n, sorted, label);
    HIP_TRY(hipMemcpyAsync(v->pos, sorted.data(), W * n * sizeof(double),
                           hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(v->label, label.data(), W * n * sizeof(unsigned short),
                           hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemsetAsync(v->ecarry, 0, W * sizeof(double), e->stream));
    int rc = qmc_evaluate_dev(e, v->W, v->pos, v->wf, nullptr, nullptr,
                              nullptr);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    v->step = 0;
    v->yield_initial = 1;
    v->tape_used = 0;
    return 0;
}

extern "C" int qmc_vmc_get_state(qmc_vmc *v, double *pos, double *wf,
                                 double *ecarry)
{
    if (!v) return fail("qmc_vmc_get_state: null argument");
    qmc_engine *e = v->eng;
    HIP_TRY(hipSetDevice(e->device));
    const size_t W = (size_t)v->W, n = (size_t)e->dm.n;
    std::vector<double> lane_pos;
    std::vector<unsigned short> label;
    if (pos) {
        lane_pos.resize(W * n);
        label.resize(W * n);
        HIP_TRY(hipMemcpyAsync(lane_pos.data(), v->pos, W * n * sizeof(double),
                               hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipMemcpyAsync(label.data(), v->label,
                               W * n * sizeof(unsigned short),
                               hipMemcpyDeviceToHost, e->stream));
    }
    if (wf) HIP_TRY(hipMemcpyAsync(wf, v->wf, W * sizeof(double),
                                   hipMemcpyDeviceToHost, e->stream));
    if (ecarry) HIP_TRY(hipMemcpyAsync(ecarry, v->ecarry, W * sizeof(double),
                                       hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (pos)       // back to the caller's particle order
        for (size_t w = 0; w < W; ++w)
            for (size_t i = 0; i < n; ++i)
                pos[w * n + label[w * n + i]] = lane_pos[w * n + i];
    return 0;
}

extern "C" int qmc_vmc_set_tape(qmc_vmc *v, const double *tape, int64_t steps)
{
    if (!v) return fail("qmc_vmc_set_tape: null argument");
    qmc_engine *e = v->eng;
    HIP_TRY(hipSetDevice(e->device));
    if (v->tape) { hipFree(v->tape); v->tape = nullptr; }
    v->tape_steps = 0;
    v->tape_used = 0;
    if (!tape || steps <= 0) return 0;
    size_t cnt = (size_t)v->W * (size_t)steps * (size_t)(e->dm.n + 1);
    if (dev_alloc(&v->tape, cnt)) return 1;
    HIP_TRY(hipMemcpy(v->tape, tape, cnt * sizeof(double),
                      hipMemcpyHostToDevice));
    v->tape_steps = steps;
    return 0;
}

extern "C" int qmc_vmc_state_dev(qmc_vmc *v, double **pos, double **wf)
{
    if (!v) return fail("null argument");
    if (pos) *pos = v->pos;
    if (wf) *wf = v->wf;
    return 0;
}

// Static structure factor parts of the current configurations, summed over
// the chains: out[m] = sum_w (|rho_m|^2, Re rho_m, Im rho_m), m < num_modes
// (qmc_base/jastrow/vmc.py:304-351 evaluates them per step of one chain; an
// ensemble gets them in one launch of the matrix-core kernel).
extern "C" int qmc_vmc_ssf(qmc_vmc *v, int32_t num_modes, double *out)
{
    if (!v || !out) return fail("qmc_vmc_ssf: null argument");
    if (num_modes <= 0 || num_modes > EST_MAXK)
        return fail("qmc_vmc_ssf: num_modes must be in [1, 256]");
    qmc_engine *e = v->eng;
    HIP_TRY(hipSetDevice(e->device));
    const int M = num_modes;
    if (M > v->ssf_cap) {
        if (v->ssf_partial) { hipFree(v->ssf_partial); hipFree(v->ssf_out); }
        v->ssf_partial = v->ssf_out = nullptr;
        if (dev_alloc(&v->ssf_partial, (size_t)EST_BLOCKS * M * 3) ||
            dev_alloc(&v->ssf_out, (size_t)M * 3))
            return 1;
        v->ssf_cap = M;
    }
    EstArgs a;
    a.ppos = v->pos; a.ref = nullptr; a.ctl = nullptr;
    a.aux_prev = nullptr; a.aux_act = nullptr; a.partial = v->ssf_partial;
    a.maxw = v->W; a.step_idx = 0; a.pfw = 0; a.n = e->dm.n; a.K = M;
    a.pure = 0; a.scale = 4.0 / e->dm.L;
    if (M <= 64) {
        const size_t lds = (size_t)(BLOCK / 64) * SsfShape<8>::WAVE_DOUBLES *
                           sizeof(double);
        allow_lds(dmc_ssf_mfma_kernel<8>, lds);
        hipLaunchKernelGGL(dmc_ssf_mfma_kernel<8>, dim3(EST_BLOCKS),
                           dim3(BLOCK), lds, e->stream, a);
    } else {
        const size_t lds = (size_t)(BLOCK / 64) * SsfShape<16>::WAVE_DOUBLES *
                           sizeof(double);
        allow_lds(dmc_ssf_mfma_kernel<16>, lds);
        hipLaunchKernelGGL(dmc_ssf_mfma_kernel<16>, dim3(EST_BLOCKS),
                           dim3(BLOCK), lds, e->stream, a);
    }
    hipLaunchKernelGGL(est_reduce_kernel, dim3((M * 3 + 31) / 32), dim3(256), 0,
                       e->stream, v->ssf_partial, EST_BLOCKS, M * 3, 1.0,
                       v->ssf_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, v->ssf_out, (size_t)M * 3 * sizeof(double),
                           hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return 0;
}

extern "C" int qmc_vmc_block_sums_dev(qmc_vmc *v, double **se, double **se2,
                                      int64_t **na)
{
    if (!v) return fail("null argument");
    if (se) *se = v->sum_e;
    if (se2) *se2 = v->sum_e2;
    if (na) *na = (int64_t *)v->n_acc;
    return 0;
}

extern "C" int qmc_vmc_run_block(qmc_vmc *v, int64_t nyield, double *sum_e,
                                 double *sum_e2, int64_t *n_acc,
                                 double *ser_wf, double *ser_e,
                                 uint8_t *ser_stat, double *ser_pos)
{
    if (!v) return fail("qmc_vmc_run_block: null argument");
    if (nyield <= 0) return fail("qmc_vmc_run_block: nyield must be >= 1");
    qmc_engine *e = v->eng;
    HIP_TRY(hipSetDevice(e->device));
    const size_t W = (size_t)v->W, ny = (size_t)nyield;
    const long long real = nyield - (v->yield_initial ? 1 : 0);
    if (v->tape && v->tape_used + real > v->tape_steps)
        return fail("qmc_vmc_run_block: tape exhausted");
    double *dwf = nullptr, *de = nullptr;
    unsigned char *dst = nullptr;
    double *dpos = nullptr;
    if ((ser_pos && dev_alloc(&dpos, ny * W * (size_t)e->dm.n)) ||
        (ser_wf && dev_alloc(&dwf, ny * W)) ||
        (ser_e && dev_alloc(&de, ny * W)) ||
        (ser_stat && dev_alloc(&dst, ny * W))) {
        if (dpos) hipFree(dpos);
        if (dwf) hipFree(dwf);
        if (de) hipFree(de);
        if (dst) hipFree(dst);
        return 1;
    }
    VmcArgs a;
    a.pos = v->pos; a.label = v->label; a.wf = v->wf; a.ecarry = v->ecarry;
    a.sum_e = v->sum_e; a.sum_e2 = v->sum_e2; a.n_acc = v->n_acc;
    a.ser_wf = dwf; a.ser_e = de; a.ser_stat = dst; a.ser_pos = dpos;
    a.tape = v->tape;
    a.tape_steps = v->tape_steps;
    a.W = v->W;
    a.gaussian = v->p.gaussian;
    a.chain0 = v->p.chain0;
    a.seed = v->p.rng_seed; a.move_spread = v->p.move_spread;
    for (long long y = 0; y < nyield; ++y) {
        a.y = y;
        a.forced = (y == 0 && v->yield_initial) ? 1 : 0;
        a.reset_sums = (y == 0) ? 1 : 0;
        a.step = v->step;
        a.tape_idx = v->tape_used;
        int rc = dispatch_shape<LaunchVmc>(e, a);
        if (rc) return rc;
        if (!a.forced) { v->step += 1; v->tape_used += 1; }
    }
    v->yield_initial = 0;
    bool need_sync = false;
    if (sum_e) { HIP_TRY(hipMemcpyAsync(sum_e, v->sum_e, W * sizeof(double),
                 hipMemcpyDeviceToHost, e->stream)); need_sync = true; }
    if (sum_e2) { HIP_TRY(hipMemcpyAsync(sum_e2, v->sum_e2, W * sizeof(double),
                  hipMemcpyDeviceToHost, e->stream)); need_sync = true; }
    if (n_acc) { HIP_TRY(hipMemcpyAsync(n_acc, v->n_acc, W * sizeof(int64_t),
                 hipMemcpyDeviceToHost, e->stream)); need_sync = true; }
    if (ser_wf) { HIP_TRY(hipMemcpyAsync(ser_wf, dwf, ny * W * sizeof(double),
                  hipMemcpyDeviceToHost, e->stream)); need_sync = true; }
    if (ser_e) { HIP_TRY(hipMemcpyAsync(ser_e, de, ny * W * sizeof(double),
                 hipMemcpyDeviceToHost, e->stream)); need_sync = true; }
    if (ser_stat) { HIP_TRY(hipMemcpyAsync(ser_stat, dst, ny * W,
                    hipMemcpyDeviceToHost, e->stream)); need_sync = true; }
    if (ser_pos) { HIP_TRY(hipMemcpyAsync(ser_pos, dpos,
                   ny * W * (size_t)e->dm.n * sizeof(double),
                   hipMemcpyDeviceToHost, e->stream)); need_sync = true; }
    if (need_sync) HIP_TRY(hipStreamSynchronize(e->stream));
    if (dpos) hipFree(dpos);
    if (dwf) hipFree(dwf);
    if (de) hipFree(de);
    if (dst) hipFree(dst);
    return 0;
}

// ----------------------------------------------------------------- DMC ----
struct qmc_dmc {
    qmc_engine *eng = nullptr;
    qmc_dmc_params p;
    long long maxw = 0;
    int nblocks = 0;
    // two population buffers: [0]/[1] alternate parent / child roles
    double *pos[2] = { nullptr, nullptr }, *drift[2] = { nullptr, nullptr };
    unsigned short *label[2] = { nullptr, nullptr };
    double *energy[2] = { nullptr, nullptr }, *weight[2] = { nullptr, nullptr };
    int cur = 0;                 // index of the parent buffer
    double *eslot = nullptr;
    double *spare = nullptr;
    long long *ref = nullptr;
    int *count = nullptr;
    long long *block_tot = nullptr, *block_off = nullptr;
    double *block_esum = nullptr;
    DmcCtl *ctl = nullptr;
    // per-step series (device), capacity ser_cap steps
    double *ser_e = nullptr, *ser_w = nullptr, *ser_ref = nullptr,
           *ser_acc = nullptr;
    unsigned long long *ser_nw = nullptr;
    long long ser_cap = 0, ser_len = 0;
    // tapes (test only)
    double *u_tape = nullptr, *g_tape = nullptr;
    std::vector<long long> u_off, g_off;
    long long tape_step = 0;
    bool stepped = false;        // a step has run since the last set_state
    std::vector<double> init_weight;   // weights of set_full_state as given (the
                                       // device keeps their logarithms)
    bool sums_pending = false;   // E_t partials still to be summed (by finish)
    double global_target = 0.0;
    // estimators (f1)
    qmc_dmc_est_params est;
    bool have_est = false;
    double *ssf_aux[2] = { nullptr, nullptr };   // [maxw][M][3]
    double *dens_aux[2] = { nullptr, nullptr };  // [maxw][B]
    double *est_partial = nullptr;               // [EST_BLOCKS][max(3M, B)]
    double *iter_ssf = nullptr, *iter_dens = nullptr;
    size_t iter_ssf_cap = 0, iter_dens_cap = 0;   // capacities in doubles
    long long est_block_steps = 0;  // steps of the estimator block in progress
    int est_last_act = 1;           // aux buffer the last estimator step wrote
};

static int dmc_reserve_series(qmc_dmc *d, long long nsteps)
{
    if (nsteps <= d->ser_cap) return 0;
    if (d->ser_e) hipFree(d->ser_e);
    if (d->ser_w) hipFree(d->ser_w);
    if (d->ser_ref) hipFree(d->ser_ref);
    if (d->ser_acc) hipFree(d->ser_acc);
    if (d->ser_nw) hipFree(d->ser_nw);
    d->ser_e = d->ser_w = d->ser_ref = d->ser_acc = nullptr;
    d->ser_nw = nullptr;
    d->ser_cap = 0;
    if (dev_alloc(&d->ser_e, nsteps) || dev_alloc(&d->ser_w, nsteps) ||
        dev_alloc(&d->ser_ref, nsteps) || dev_alloc(&d->ser_acc, nsteps) ||
        dev_alloc(&d->ser_nw, nsteps)) {
        // dev_alloc nulls what it could not allocate
        if (d->ser_e) hipFree(d->ser_e);
        if (d->ser_w) hipFree(d->ser_w);
        if (d->ser_ref) hipFree(d->ser_ref);
        if (d->ser_acc) hipFree(d->ser_acc);
        if (d->ser_nw) hipFree(d->ser_nw);
        d->ser_e = d->ser_w = d->ser_ref = d->ser_acc = nullptr;
        d->ser_nw = nullptr;
        return 1;
    }
    d->ser_cap = nsteps;
    return 0;
}

extern "C" void qmc_dmc_destroy(qmc_dmc *d);

extern "C" int qmc_dmc_create(qmc_engine *e, const qmc_dmc_params *p,
                              qmc_dmc **out)
{
    if (!e || !p || !out) return fail("qmc_dmc_create: null argument");
    if (p->max_num_walkers <= 0 || p->target_num_walkers <= 0)
        return fail("qmc_dmc_create: walker counts must be positive");
    if (!(p->time_step > 0)) return fail("qmc_dmc_create: time_step <= 0");
    HIP_TRY(hipSetDevice(e->device));
    qmc_dmc *d = new qmc_dmc();
    d->eng = e;
    d->p = *p;
    d->maxw = p->max_num_walkers;
    d->nblocks = (int)((d->maxw + BR_TILE - 1) / BR_TILE);
    d->global_target = (double)p->target_num_walkers;
    const size_t W = (size_t)d->maxw, n = (size_t)e->dm.n;
    int rc = 0;
    for (int b = 0; b < 2 && !rc; ++b) {
        rc |= dev_alloc(&d->pos[b], W * n) || dev_alloc(&d->drift[b], W * n) ||
              dev_alloc(&d->label[b], W * n) ||
              dev_alloc(&d->energy[b], W) || dev_alloc(&d->weight[b], W);
    }
    // (the cached second Box-Muller normal: one particle per lane only,
    // qmc_kernels.h: DmcSpare)
    const bool need_spare = QMC_DMC_SPARE && e->P == 1;
    rc = rc || dev_alloc(&d->eslot, W) ||
         (need_spare && dev_alloc(&d->spare, W * n)) ||
         dev_alloc(&d->ref, W) ||
         dev_alloc(&d->count, W) || dev_alloc(&d->block_tot, d->nblocks) ||
         dev_alloc(&d->block_off, d->nblocks) ||
         dev_alloc(&d->block_esum, d->nblocks) || dev_alloc(&d->ctl, 1);
    if (rc) { qmc_dmc_destroy(d); return 1; }
    HIP_TRY(hipMemset(d->ctl, 0, sizeof(DmcCtl)));
    HIP_TRY(hipMemset(d->ref, 0, W * sizeof(long long)));
    HIP_TRY(hipMemset(d->eslot, 0, W * sizeof(double)));
    *out = d;
    return 0;
}

extern "C" void qmc_dmc_destroy(qmc_dmc *d)
{
    if (!d) return;
    hipSetDevice(d->eng->device);
    // (hipFree(nullptr) is a no-op: a partly built ensemble is fine here)
    for (int b = 0; b < 2; ++b) {
        hipFree(d->pos[b]); hipFree(d->drift[b]); hipFree(d->label[b]);
        hipFree(d->energy[b]); hipFree(d->weight[b]);
    }
    hipFree(d->eslot); hipFree(d->spare); hipFree(d->ref); hipFree(d->count);
    hipFree(d->block_tot); hipFree(d->block_off); hipFree(d->block_esum);
    hipFree(d->ctl);
    if (d->ser_e) { hipFree(d->ser_e); hipFree(d->ser_w); hipFree(d->ser_ref);
                    hipFree(d->ser_acc); hipFree(d->ser_nw); }
    if (d->u_tape) hipFree(d->u_tape);
    if (d->g_tape) hipFree(d->g_tape);
    for (int k = 0; k < 2; ++k) {
        if (d->ssf_aux[k]) hipFree(d->ssf_aux[k]);
        if (d->dens_aux[k]) hipFree(d->dens_aux[k]);
    }
    if (d->est_partial) hipFree(d->est_partial);
    if (d->iter_ssf) hipFree(d->iter_ssf);
    if (d->iter_dens) hipFree(d->iter_dens);
    delete d;
}

static int dmc_reset_ctl(qmc_dmc *d, long long nw, double ref_energy)
{
    DmcCtl c;
    memset(&c, 0, sizeof(c));
    c.prev_nw = nw;
    c.nw = nw;
    c.ref_energy = ref_energy;
    HIP_TRY(hipMemcpyAsync(d->ctl, &c, sizeof(c), hipMemcpyHostToDevice,
                           d->eng->stream));
    HIP_TRY(hipStreamSynchronize(d->eng->stream));
    d->cur = 0;
    d->stepped = false;
    d->tape_step = 0;
    d->ser_len = 0;
    return 0;
}

static int dmc_zero_population(qmc_dmc *d)
{
    d->init_weight.clear();
    qmc_engine *e = d->eng;
    const size_t n = (size_t)e->dm.n, W = (size_t)d->maxw;
    for (int b = 0; b < 2; ++b) {
        HIP_TRY(hipMemsetAsync(d->pos[b], 0, W * n * sizeof(double), e->stream));
        HIP_TRY(hipMemsetAsync(d->drift[b], 0, W * n * sizeof(double), e->stream));
        HIP_TRY(hipMemsetAsync(d->energy[b], 0, W * sizeof(double), e->stream));
        HIP_TRY(hipMemsetAsync(d->weight[b], 0, W * sizeof(double), e->stream));
        hipLaunchKernelGGL(ident_labels_kernel,
                           dim3((unsigned)((W * n + 255) / 256)), dim3(256), 0,
                           e->stream, d->label[b], (long long)W, (int)n);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemsetAsync(d->eslot, 0, W * sizeof(double), e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return 0;
}

// pos: host positions (sorted here) or device positions already in lane order
// with their labels (label_dev, may be null = identity).
static int dmc_set_state_impl(qmc_dmc *d, int64_t nw, const double *pos,
                              bool pos_on_device,
                              const unsigned short *label_dev, int use_ref,
                              double ref_energy, int64_t src_rows = 0)
{
    if (!d || !pos) return fail("qmc_dmc_set_state: null argument");
    if (nw <= 0 || nw > d->maxw)
        return fail("qmc_dmc_set_state: number of walkers out of range");
    qmc_engine *e = d->eng;
    HIP_TRY(hipSetDevice(e->device));
    const size_t n = (size_t)e->dm.n;
    if (dmc_zero_population(d)) return 1;
    if (pos_on_device && src_rows > 0 && src_rows < nw) {
        // more walkers than source rows: the rows are reused cyclically
        const long long tot = (long long)nw * (long long)n;
        const unsigned grid = (unsigned)((tot + 255) / 256);
        hipLaunchKernelGGL(tile_rows_kernel<double>, dim3(grid), dim3(256), 0,
                           e->stream, pos, (long long)src_rows, d->pos[0],
                           (long long)nw, (int)n);
        if (label_dev)
            hipLaunchKernelGGL(tile_rows_kernel<unsigned short>, dim3(grid),
                               dim3(256), 0, e->stream, label_dev,
                               (long long)src_rows, d->label[0], (long long)nw,
                               (int)n);
        HIP_TRY(hipGetLastError());
    } else if (pos_on_device) {
        HIP_TRY(hipMemcpyAsync(d->pos[0], pos, (size_t)nw * n * sizeof(double),
                               hipMemcpyDeviceToDevice, e->stream));
        if (label_dev)
            HIP_TRY(hipMemcpyAsync(d->label[0], label_dev,
                                   (size_t)nw * n * sizeof(unsigned short),
                                   hipMemcpyDeviceToDevice, e->stream));
    } else {
        std::vector<double> sorted;
        std::vector<unsigned short> label;
        sort_rows(pos, (size_t)nw, n, sorted, label);
        HIP_TRY(hipMemcpyAsync(d->pos[0], sorted.data(),
                               (size_t)nw * n * sizeof(double),
                               hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipMemcpyAsync(d->label[0], label.data(),
                               (size_t)nw * n * sizeof(unsigned short),
                               hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    PrepArgs a{ d->pos[0], d->drift[0], d->energy[0], (long long)nw };
    int rc = dispatch_shape<LaunchPrep>(e, a);
    if (rc) return rc;
    std::vector<double> en((size_t)nw);
    // unit weights; the device keeps LOG weights (dmc_evolve_kernel)
    HIP_TRY(hipMemsetAsync(d->weight[0], 0, (size_t)nw * sizeof(double), e->stream));
    HIP_TRY(hipMemcpyAsync(en.data(), d->energy[0], (size_t)nw * sizeof(double),
                           hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(d->eslot, d->energy[0], (size_t)nw * sizeof(double),
                           hipMemcpyDeviceToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (!use_ref) {
        // mrbp_qmc/dmc.py:302-312: sum(E w) / sum(w) with unit weights
        double se = 0.0;
        for (size_t i = 0; i < (size_t)nw; ++i) se += en[i] * 1.0;
        ref_energy = se / (double)nw;
    }
    return dmc_reset_ctl(d, nw, ref_energy);
}

extern "C" int qmc_dmc_set_state(qmc_dmc *d, int64_t nw, const double *pos,
                                 int use_ref, double ref_energy)
{
    return dmc_set_state_impl(d, nw, pos, false, nullptr, use_ref, ref_energy);
}

extern "C" int qmc_dmc_set_state_dev(qmc_dmc *d, int64_t nw,
                                     const double *pos_dev, int use_ref,
                                     double ref_energy)
{
    return dmc_set_state_impl(d, nw, pos_dev, true, nullptr, use_ref,
                              ref_energy);
}

extern "C" int qmc_dmc_set_state_from_vmc(qmc_dmc *d, qmc_vmc *v, int64_t nw,
                                          int use_ref, double ref_energy)
{
    if (!d || !v) return fail("qmc_dmc_set_state_from_vmc: null argument");
    if (v->eng != d->eng)
        return fail("qmc_dmc_set_state_from_vmc: ensembles of different engines");
    // nw > num_chains: the chains are reused cyclically
    return dmc_set_state_impl(d, nw, v->pos, true, v->label, use_ref,
                              ref_energy, v->W);
}

extern "C" int qmc_dmc_set_full_state(qmc_dmc *d, int64_t nw,
                                      const double *confs,
                                      const double *energy,
                                      const double *weight,
                                      const double *slot_energy,
                                      double ref_energy)
{
    if (!d || !confs || !energy || !weight)
        return fail("qmc_dmc_set_full_state: null argument");
    if (nw <= 0 || nw > d->maxw)
        return fail("qmc_dmc_set_full_state: number of walkers out of range");
    qmc_engine *e = d->eng;
    HIP_TRY(hipSetDevice(e->device));
    const size_t n = (size_t)e->dm.n, W = (size_t)d->maxw;
    // confs[s] = (pos row, drift row): sort by position, carry the drift along
    std::vector<double> hp, hd;
    std::vector<unsigned short> label;
    sort_rows(confs, (size_t)nw, n, hp, label, confs + n, &hd, 2 * n, 2 * n);
    if (dmc_zero_population(d)) return 1;
    HIP_TRY(hipMemcpyAsync(d->pos[0], hp.data(), hp.size() * sizeof(double),
                           hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(d->drift[0], hd.data(), hd.size() * sizeof(double),
                           hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(d->label[0], label.data(),
                           label.size() * sizeof(unsigned short),
                           hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(d->energy[0], energy, (size_t)nw * sizeof(double),
                           hipMemcpyHostToDevice, e->stream));
    // the device keeps LOG weights (dmc_evolve_kernel); a weight <= 0 has no
    // children either way
    std::vector<double> logw((size_t)nw);
    d->init_weight.assign(weight, weight + nw);
    for (size_t i = 0; i < (size_t)nw; ++i)
        logw[i] = weight[i] > 0.0 ? std::log(weight[i]) : -HUGE_VAL;
    HIP_TRY(hipMemcpyAsync(d->weight[0], logw.data(), (size_t)nw * sizeof(double),
                           hipMemcpyHostToDevice, e->stream));
    // the reference copies the whole props.energy array of the initial state
    // into its `actual` buffer (qmc_base/dmc.py:707-708), stale tail included
    if (slot_energy)
        HIP_TRY(hipMemcpyAsync(d->eslot, slot_energy, W * sizeof(double),
                               hipMemcpyHostToDevice, e->stream));
    else
        HIP_TRY(hipMemcpyAsync(d->eslot, energy, (size_t)nw * sizeof(double),
                               hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return dmc_reset_ctl(d, nw, ref_energy);
}

extern "C" int qmc_dmc_set_tape(qmc_dmc *d, const double *u, int64_t nu,
                                const double *g, int64_t ng,
                                const int64_t *u_off, const int64_t *g_off,
                                int64_t nsteps)
{
    if (!d) return fail("qmc_dmc_set_tape: null argument");
    HIP_TRY(hipSetDevice(d->eng->device));
    if (d->u_tape) { hipFree(d->u_tape); d->u_tape = nullptr; }
    if (d->g_tape) { hipFree(d->g_tape); d->g_tape = nullptr; }
    d->u_off.clear(); d->g_off.clear();
    d->tape_step = 0;
    if (!u || !g || nsteps <= 0) return 0;
    // pad so that a step may read up to maxw uniforms / maxw*N normals
    size_t upad = (size_t)nu + (size_t)d->maxw;
    size_t gpad = (size_t)ng + (size_t)d->maxw * (size_t)d->eng->dm.n;
    if (dev_alloc(&d->u_tape, upad) || dev_alloc(&d->g_tape, gpad)) return 1;
    HIP_TRY(hipMemset(d->u_tape, 0, upad * sizeof(double)));
    HIP_TRY(hipMemset(d->g_tape, 0, gpad * sizeof(double)));
    HIP_TRY(hipMemcpy(d->u_tape, u, (size_t)nu * sizeof(double),
                      hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d->g_tape, g, (size_t)ng * sizeof(double),
                      hipMemcpyHostToDevice));
    d->u_off.assign(u_off, u_off + nsteps);
    d->g_off.assign(g_off, g_off + nsteps);
    return 0;
}

static FinishArgs dmc_finish_args(qmc_dmc *d, const double *total_dev,
                                  long long ser_idx)
{
    FinishArgs f;
    f.ctl = d->ctl; f.total = total_dev;
    f.block_esum = (!total_dev && d->sums_pending) ? d->block_esum : nullptr;
    const bool rec = ser_idx >= 0;
    f.ser_e = rec ? d->ser_e : nullptr; f.ser_w = d->ser_w;
    f.ser_ref = d->ser_ref; f.ser_acc = d->ser_acc; f.ser_nw = d->ser_nw;
    f.ser_idx = ser_idx;
    f.kappa = d->p.num_walkers_control_factor; f.dt = d->p.time_step;
    f.target = d->global_target;
    return f;
}

// Small populations run the whole branching step in one workgroup
// (branch_fused_kernel), which can also carry the previous step's bookkeeping.
static bool dmc_fused_branching(const qmc_dmc *d)
{
    return d->nblocks <= BR_FUSED_TILES;
}

// Enqueue the rank-local part of one time step.  `prev_fin`: the bookkeeping of
// the step before, to ride at the head of the fused branching kernel (only
// where dmc_fused_branching(d)).
// `prev_fin_done` (if given) is set once the fused branching kernel -- which
// applies the previous step's deferred bookkeeping `prev_fin` at its head --
// has been enqueued: a caller that sees an error must launch that bookkeeping
// itself only while the flag is still false.
static int dmc_enqueue_local(qmc_dmc *d, double *partial_dev,
                             const FinishArgs *prev_fin = nullptr,
                             bool *prev_fin_done = nullptr)
{
    qmc_engine *e = d->eng;
    const int par = d->cur, chi = 1 - d->cur;
    const double *ut = nullptr, *gt = nullptr;
    if (d->u_tape) {
        if (d->tape_step >= (long long)d->u_off.size())
            return fail("qmc_dmc: tape exhausted");
        ut = d->u_tape + d->u_off[(size_t)d->tape_step];
        gt = d->g_tape + d->g_off[(size_t)d->tape_step];
        d->tape_step += 1;
    }
    BranchArgs b;
    b.weight = d->weight[par]; b.energy = d->energy[par];
    b.count = d->count; b.block_tot = d->block_tot; b.block_off = d->block_off;
    b.block_esum = d->block_esum; b.ref = d->ref; b.ctl = d->ctl;
    b.u_tape = ut; b.maxw = d->maxw; b.seed = d->p.rng_seed;
    b.slot0 = d->p.slot0;
    if (dmc_fused_branching(d)) {
        d->sums_pending = false;
        FinishArgs none{};
        hipLaunchKernelGGL(branch_fused_kernel, dim3(1), dim3(BLOCK), 0,
                           e->stream, b, partial_dev,
                           prev_fin ? *prev_fin : none, prev_fin ? 1 : 0);
        if (prev_fin_done) *prev_fin_done = true;
    } else {
        if (prev_fin) return fail("qmc_dmc: internal: deferred bookkeeping "
                                  "needs the fused branching kernel");
        hipLaunchKernelGGL(branch_count_kernel, dim3(d->nblocks), dim3(BLOCK),
                           0, e->stream, b);
        hipLaunchKernelGGL(branch_scatter_kernel, dim3(d->nblocks),
                           dim3(BLOCK), 0, e->stream, b);
        // single GPU: the finish kernel sums the partials itself
        d->sums_pending = partial_dev == nullptr;
        if (partial_dev)
            hipLaunchKernelGGL(dmc_local_sums_kernel, dim3(1), dim3(BLOCK), 0,
                               e->stream, d->block_esum, d->ctl, partial_dev);
    }
    HIP_TRY(hipGetLastError());
    EvolveArgs a;
    a.ppos = d->pos[par]; a.pdrift = d->drift[par]; a.penergy = d->energy[par];
    a.cpos = d->pos[chi]; a.cdrift = d->drift[chi];
    a.plabel = d->label[par]; a.clabel = d->label[chi];
    a.cenergy = d->energy[chi]; a.cweight = d->weight[chi];
    a.eslot = d->eslot; a.ref = d->ref; a.ctl = d->ctl; a.g_tape = gt;
    a.spare = d->spare;
    a.maxw = d->maxw; a.dt = d->p.time_step;
    a.sigma = sqrt(2 * d->p.time_step);           // mrbp_qmc/dmc.py:178
    a.seed = d->p.rng_seed; a.slot0 = d->p.slot0;
    a.fix_stale = d->p.fix_stale_energy;
    int rc = dispatch_shape<LaunchEvolve>(e, a);
    if (rc) return rc;
    return 0;
}

static int dmc_enqueue_finish(qmc_dmc *d, const double *total_dev,
                              long long ser_idx)
{
    qmc_engine *e = d->eng;
    const FinishArgs f = dmc_finish_args(d, total_dev, ser_idx);
    hipLaunchKernelGGL(dmc_finish_kernel, dim3(1), dim3(BLOCK), 0, e->stream,
                       f);
    HIP_TRY(hipGetLastError());
    d->cur = 1 - d->cur;          // children become the parents
    d->stepped = true;
    return 0;
}

extern "C" int qmc_dmc_read_series(qmc_dmc *d, int64_t nsteps, double *energy,
                                   double *weight, uint64_t *num_walkers,
                                   double *ref_energy, double *accum_energy)
{
    if (!d) return fail("qmc_dmc_read_series: null argument");
    if (nsteps > d->ser_len) return fail("qmc_dmc_read_series: too many steps");
    qmc_engine *e = d->eng;
    HIP_TRY(hipSetDevice(e->device));
    const size_t ns = (size_t)nsteps;
    if (energy) HIP_TRY(hipMemcpyAsync(energy, d->ser_e, ns * 8,
                                       hipMemcpyDeviceToHost, e->stream));
    if (weight) HIP_TRY(hipMemcpyAsync(weight, d->ser_w, ns * 8,
                                       hipMemcpyDeviceToHost, e->stream));
    if (num_walkers) HIP_TRY(hipMemcpyAsync(num_walkers, d->ser_nw, ns * 8,
                                            hipMemcpyDeviceToHost, e->stream));
    if (ref_energy) HIP_TRY(hipMemcpyAsync(ref_energy, d->ser_ref, ns * 8,
                                           hipMemcpyDeviceToHost, e->stream));
    if (accum_energy) HIP_TRY(hipMemcpyAsync(accum_energy, d->ser_acc, ns * 8,
                                             hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    // a full read consumes the series (split-step drivers append to it)
    if (nsteps == d->ser_len) d->ser_len = 0;
    return 0;
}

extern "C" int qmc_dmc_run_block(qmc_dmc *d, int64_t nsteps, double *energy,
                                 double *weight, uint64_t *num_walkers,
                                 double *ref_energy, double *accum_energy)
{
    if (!d) return fail("qmc_dmc_run_block: null argument");
    if (nsteps <= 0) return fail("qmc_dmc_run_block: nsteps must be >= 1");
    if (d->p.external_reduce)
        return fail("qmc_dmc_run_block: ensemble was created for "
                    "external_reduce; drive it with step_local/step_finish");
    HIP_TRY(hipSetDevice(d->eng->device));
    if (dmc_reserve_series(d, nsteps)) return 1;
    d->ser_len = 0;
    if (dmc_fused_branching(d)) {
        // small population: a step costs the latency of its dependent
        // launches.  The bookkeeping of step t rides at the head of step
        // t + 1's branching kernel (same order on the stream as its own
        // launch would have); only the last step of the block launches it.
        FinishArgs pend{};
        bool have = false;
        for (long long t = 0; t < nsteps; ++t) {
            bool pend_done = false;
            int rc = dmc_enqueue_local(d, nullptr, have ? &pend : nullptr,
                                       &pend_done);
            if (rc) {
                // the step before still needs its bookkeeping -- unless this
                // step's branching kernel, which carries it, was enqueued
                // before the failure (then it has been applied, once)
                if (have && !pend_done) {
                    hipLaunchKernelGGL(dmc_finish_kernel, dim3(1), dim3(BLOCK),
                                       0, d->eng->stream, pend);
                    (void)hipGetLastError();   // the first error is reported
                }
                return rc;
            }
            pend = dmc_finish_args(d, nullptr, t);
            have = true;
            d->cur = 1 - d->cur;          // children become the parents
            d->stepped = true;
            d->ser_len = t + 1;
        }
        hipLaunchKernelGGL(dmc_finish_kernel, dim3(1), dim3(BLOCK), 0,
                           d->eng->stream, pend);
        HIP_TRY(hipGetLastError());
    } else {
        for (long long t = 0; t < nsteps; ++t) {
            int rc = dmc_enqueue_local(d, nullptr);
            if (rc) return rc;
            rc = dmc_enqueue_finish(d, nullptr, t);
            if (rc) return rc;
            d->ser_len = t + 1;
        }
    }
    if (energy || weight || num_walkers || ref_energy || accum_energy)
        return qmc_dmc_read_series(d, nsteps, energy, weight, num_walkers,
                                   ref_energy, accum_energy);
    return 0;
}

extern "C" int qmc_dmc_set_estimators(qmc_dmc *d, const qmc_dmc_est_params *p)
{
    if (!d || !p) return fail("qmc_dmc_set_estimators: null argument");
    if (p->num_modes < 0 || p->num_modes > EST_MAXK || p->num_bins < 0 ||
        p->num_bins > EST_MAXK)
        return fail("qmc_dmc_set_estimators: num_modes / num_bins must be in "
                    "[0, 256]");
    qmc_engine *e = d->eng;
    HIP_TRY(hipSetDevice(e->device));
    for (int k = 0; k < 2; ++k) {
        if (d->ssf_aux[k]) { hipFree(d->ssf_aux[k]); d->ssf_aux[k] = nullptr; }
        if (d->dens_aux[k]) { hipFree(d->dens_aux[k]); d->dens_aux[k] = nullptr; }
    }
    if (d->est_partial) { hipFree(d->est_partial); d->est_partial = nullptr; }
    // the per-step output buffers are sized for the old mode / bin counts
    if (d->iter_ssf) { hipFree(d->iter_ssf); d->iter_ssf = nullptr; }
    if (d->iter_dens) { hipFree(d->iter_dens); d->iter_dens = nullptr; }
    d->iter_ssf_cap = d->iter_dens_cap = 0;
    d->est_block_steps = 0;
    d->est_last_act = 1;
    d->have_est = false;
    d->est = *p;
    const bool want = p->num_modes > 0 || p->num_bins > 0;
    const size_t W = (size_t)d->maxw;
    // estimators are on only once every buffer exists: a failed allocation
    // leaves the population without estimators (and without half of their
    // buffers), not with `have_est` set over null pointers
    bool failed = false;
    if (p->num_modes > 0)
        for (int k = 0; k < 2 && !failed; ++k)
            failed = dev_alloc(&d->ssf_aux[k], W * (size_t)p->num_modes * 3) != 0;
    if (p->num_bins > 0)
        for (int k = 0; k < 2 && !failed; ++k)
            failed = dev_alloc(&d->dens_aux[k], W * (size_t)p->num_bins) != 0;
    size_t kc = (size_t)(p->num_modes * 3 > p->num_bins ? p->num_modes * 3
                                                        : p->num_bins);
    if (want && !failed)
        failed = dev_alloc(&d->est_partial, EST_BLOCKS * kc) != 0;
    if (failed) {
        for (int k = 0; k < 2; ++k) {
            if (d->ssf_aux[k]) { hipFree(d->ssf_aux[k]); d->ssf_aux[k] = nullptr; }
            if (d->dens_aux[k]) { hipFree(d->dens_aux[k]); d->dens_aux[k] = nullptr; }
        }
        if (d->est_partial) { hipFree(d->est_partial); d->est_partial = nullptr; }
        d->est.num_modes = d->est.num_bins = 0;
        return 1;            // (dev_alloc has set the message)
    }
    d->have_est = want;
    return 0;
}

// Evaluate the estimators on the population yielded by the step that has just
// been finished (its parents are the buffer that is not `cur`).
static int dmc_enqueue_estimators(qmc_dmc *d, long long step_idx)
{
    qmc_engine *e = d->eng;
    const int par = 1 - d->cur;
    const int act = (int)(step_idx % 2), prev = 1 - act;
    d->est_last_act = act;
    EstArgs a;
    a.ppos = d->pos[par]; a.ref = d->ref; a.ctl = d->ctl;
    a.maxw = d->maxw; a.step_idx = step_idx; a.n = e->dm.n;
    a.partial = d->est_partial;
    if (d->est.num_modes > 0) {
        const int M = d->est.num_modes;
        a.aux_prev = d->ssf_aux[prev]; a.aux_act = d->ssf_aux[act];
        a.K = M; a.pure = d->est.ssf_pure; a.pfw = d->est.ssf_pfw;
        a.scale = 4.0 / e->dm.L;
        if (M <= 64) {
            const size_t lds = (size_t)(BLOCK / 64) *
                               SsfShape<8>::WAVE_DOUBLES * sizeof(double);
            allow_lds(dmc_ssf_mfma_kernel<8>, lds);
            hipLaunchKernelGGL(dmc_ssf_mfma_kernel<8>, dim3(EST_BLOCKS),
                               dim3(BLOCK), lds, e->stream, a);
        } else {
            const size_t lds = (size_t)(BLOCK / 64) *
                               SsfShape<16>::WAVE_DOUBLES * sizeof(double);
            allow_lds(dmc_ssf_mfma_kernel<16>, lds);
            hipLaunchKernelGGL(dmc_ssf_mfma_kernel<16>, dim3(EST_BLOCKS),
                               dim3(BLOCK), lds, e->stream, a);
        }
        double div = 1.0;
        if (a.pure) div = step_idx < a.pfw ? (double)(step_idx + 1)
                                           : (double)a.pfw;
        hipLaunchKernelGGL(est_reduce_kernel, dim3((M * 3 + 31) / 32),
                           dim3(256), 0, e->stream, d->est_partial, EST_BLOCKS,
                           M * 3, div, d->iter_ssf + (size_t)step_idx * M * 3);
    }
    if (d->est.num_bins > 0) {
        const int B = d->est.num_bins;
        a.aux_prev = d->dens_aux[prev]; a.aux_act = d->dens_aux[act];
        a.K = B; a.pure = d->est.dens_pure; a.pfw = d->est.dens_pfw;
        a.scale = e->dm.L / (double)B;      // bin size
        hipLaunchKernelGGL(dmc_density_kernel, dim3(EST_BLOCKS), dim3(BLOCK), 0,
                           e->stream, a);
        double div = 1.0;
        if (a.pure) div = step_idx < a.pfw ? (double)(step_idx + 1)
                                           : (double)a.pfw;
        hipLaunchKernelGGL(est_reduce_kernel, dim3((B + 31) / 32), dim3(256),
                           0, e->stream, d->est_partial, EST_BLOCKS, B, div,
                           d->iter_dens + (size_t)step_idx * B);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// Start an estimator block of `nsteps` steps: per-block resets of the
// forward-walking buffers and of the per-step outputs (qmc_base/dmc.py:897-909).
static int dmc_est_begin_block(qmc_dmc *d, long long nsteps)
{
    qmc_engine *e = d->eng;
    const size_t M3 = (size_t)d->est.num_modes * 3, B = (size_t)d->est.num_bins;
    const size_t W = (size_t)d->maxw;
    const size_t need_s = (size_t)nsteps * (M3 ? M3 : 1);
    const size_t need_d = (size_t)nsteps * (B ? B : 1);
    if (need_s > d->iter_ssf_cap) {
        if (d->iter_ssf) { hipFree(d->iter_ssf); d->iter_ssf = nullptr; }
        d->iter_ssf_cap = 0;
        if (dev_alloc(&d->iter_ssf, need_s)) return 1;
        d->iter_ssf_cap = need_s;
    }
    if (need_d > d->iter_dens_cap) {
        if (d->iter_dens) { hipFree(d->iter_dens); d->iter_dens = nullptr; }
        d->iter_dens_cap = 0;
        if (dev_alloc(&d->iter_dens, need_d)) return 1;
        d->iter_dens_cap = need_d;
    }
    HIP_TRY(hipMemsetAsync(d->iter_ssf, 0, need_s * 8, e->stream));
    HIP_TRY(hipMemsetAsync(d->iter_dens, 0, need_d * 8, e->stream));
    for (int k = 0; k < 2; ++k) {
        if (M3) HIP_TRY(hipMemsetAsync(d->ssf_aux[k], 0, W * M3 * 8, e->stream));
        if (B) HIP_TRY(hipMemsetAsync(d->dens_aux[k], 0, W * B * 8, e->stream));
    }
    d->est_block_steps = nsteps;
    d->est_last_act = 1;
    return 0;
}

extern "C" int qmc_dmc_run_block_est(qmc_dmc *d, int64_t nsteps,
                                     int eval_estimators, double *energy,
                                     double *weight, uint64_t *num_walkers,
                                     double *ref_energy, double *accum_energy,
                                     double *iter_ssf, double *iter_density)
{
    if (!d) return fail("qmc_dmc_run_block_est: null argument");
    if (nsteps <= 0) return fail("qmc_dmc_run_block_est: nsteps must be >= 1");
    if (d->p.external_reduce)
        return fail("qmc_dmc_run_block_est: ensemble was created for "
                    "external_reduce; drive it with est_begin_block / "
                    "step_local / step_finish / step_estimators");
    if (!d->have_est)
        return qmc_dmc_run_block(d, nsteps, energy, weight, num_walkers,
                                 ref_energy, accum_energy);
    qmc_engine *e = d->eng;
    HIP_TRY(hipSetDevice(e->device));
    if (dmc_reserve_series(d, nsteps)) return 1;
    if (dmc_est_begin_block(d, nsteps)) return 1;
    const size_t M3 = (size_t)d->est.num_modes * 3, B = (size_t)d->est.num_bins;
    d->ser_len = 0;
    for (long long t = 0; t < nsteps; ++t) {
        int rc = dmc_enqueue_local(d, nullptr);
        if (rc) return rc;
        rc = dmc_enqueue_finish(d, nullptr, t);
        if (rc) return rc;
        d->ser_len = t + 1;
        if (eval_estimators) {
            rc = dmc_enqueue_estimators(d, t);
            if (rc) return rc;
        }
    }
    if (iter_ssf && M3)
        HIP_TRY(hipMemcpyAsync(iter_ssf, d->iter_ssf, (size_t)nsteps * M3 * 8,
                               hipMemcpyDeviceToHost, e->stream));
    if (iter_density && B)
        HIP_TRY(hipMemcpyAsync(iter_density, d->iter_dens, (size_t)nsteps * B * 8,
                               hipMemcpyDeviceToHost, e->stream));
    return qmc_dmc_read_series(d, nsteps, energy, weight, num_walkers,
                               ref_energy, accum_energy);
}

// Split-step counterparts (multi-GPU, external_reduce): the caller opens an
// estimator block, calls step_estimators after every step_finish of a kept
// block, and sums the per-rank outputs (linear in the walkers) across ranks.
extern "C" int qmc_dmc_est_begin_block(qmc_dmc *d, int64_t nsteps)
{
    if (!d) return fail("qmc_dmc_est_begin_block: null argument");
    if (!d->have_est)
        return fail("qmc_dmc_est_begin_block: no estimators are set");
    if (nsteps <= 0) return fail("qmc_dmc_est_begin_block: nsteps must be >= 1");
    HIP_TRY(hipSetDevice(d->eng->device));
    return dmc_est_begin_block(d, nsteps);
}

extern "C" int qmc_dmc_step_estimators(qmc_dmc *d, int64_t step_idx)
{
    if (!d) return fail("qmc_dmc_step_estimators: null argument");
    if (!d->have_est)
        return fail("qmc_dmc_step_estimators: no estimators are set");
    if (step_idx < 0 || step_idx >= d->est_block_steps)
        return fail("qmc_dmc_step_estimators: step index outside the block "
                    "opened by qmc_dmc_est_begin_block");
    if (!d->stepped)
        return fail("qmc_dmc_step_estimators: no time step has run yet");
    HIP_TRY(hipSetDevice(d->eng->device));
    return dmc_enqueue_estimators(d, step_idx);
}

extern "C" int qmc_dmc_est_iter_dev(qmc_dmc *d, double **iter_ssf,
                                    double **iter_density)
{
    if (!d) return fail("qmc_dmc_est_iter_dev: null argument");
    if (iter_ssf) *iter_ssf = d->est.num_modes > 0 ? d->iter_ssf : nullptr;
    if (iter_density) *iter_density = d->est.num_bins > 0 ? d->iter_dens : nullptr;
    return 0;
}

extern "C" int qmc_dmc_step_local(qmc_dmc *d, double *partial_dev)
{
    if (!d || !partial_dev) return fail("qmc_dmc_step_local: null argument");
    HIP_TRY(hipSetDevice(d->eng->device));
    return dmc_enqueue_local(d, partial_dev);
}

extern "C" int qmc_dmc_step_finish(qmc_dmc *d, const double *total_dev)
{
    if (!d || !total_dev) return fail("qmc_dmc_step_finish: null argument");
    HIP_TRY(hipSetDevice(d->eng->device));
    if (d->ser_len >= d->ser_cap) {
        // grow geometrically, keeping what was recorded
        long long ncap = d->ser_cap ? d->ser_cap * 2 : 1024;
        double *oe = d->ser_e, *ow = d->ser_w, *orf = d->ser_ref,
               *oa = d->ser_acc;
        unsigned long long *on = d->ser_nw;
        long long olen = d->ser_len;
        d->ser_e = nullptr; d->ser_cap = 0;
        if (dmc_reserve_series(d, ncap)) return 1;
        if (oe) {
            hipStream_t s = d->eng->stream;
            HIP_TRY(hipMemcpyAsync(d->ser_e, oe, olen * 8, hipMemcpyDeviceToDevice, s));
            HIP_TRY(hipMemcpyAsync(d->ser_w, ow, olen * 8, hipMemcpyDeviceToDevice, s));
            HIP_TRY(hipMemcpyAsync(d->ser_ref, orf, olen * 8, hipMemcpyDeviceToDevice, s));
            HIP_TRY(hipMemcpyAsync(d->ser_acc, oa, olen * 8, hipMemcpyDeviceToDevice, s));
            HIP_TRY(hipMemcpyAsync(d->ser_nw, on, olen * 8, hipMemcpyDeviceToDevice, s));
            HIP_TRY(hipStreamSynchronize(s));
            hipFree(oe); hipFree(ow); hipFree(orf); hipFree(oa); hipFree(on);
        }
    }
    int rc = dmc_enqueue_finish(d, total_dev, d->ser_len);
    if (rc) return rc;
    d->ser_len += 1;
    return 0;
}

extern "C" int qmc_dmc_num_walkers(qmc_dmc *d, int64_t *nw)
{
    if (!d || !nw) return fail("qmc_dmc_num_walkers: null argument");
    HIP_TRY(hipSetDevice(d->eng->device));
    DmcCtl c;
    HIP_TRY(hipMemcpyAsync(&c, d->ctl, sizeof(c), hipMemcpyDeviceToHost,
                           d->eng->stream));
    HIP_TRY(hipStreamSynchronize(d->eng->stream));
    *nw = c.prev_nw;
    return 0;
}

extern "C" int qmc_dmc_get_state(qmc_dmc *d, double *confs, double *energy,
                                 double *weight, uint8_t *mask,
                                 int64_t *cloning_ref, double *scalars)
{
    if (!d) return fail("qmc_dmc_get_state: null argument");
    qmc_engine *e = d->eng;
    HIP_TRY(hipSetDevice(e->device));
    const size_t n = (size_t)e->dm.n, W = (size_t)d->maxw;
    DmcCtl c;
    HIP_TRY(hipMemcpyAsync(&c, d->ctl, sizeof(c), hipMemcpyDeviceToHost,
                           e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    const long long nw = d->stepped ? c.nw : c.prev_nw;
    std::vector<long long> href;
    if (cloning_ref) {
        href.assign(W, 0);
        if (d->stepped)
            HIP_TRY(hipMemcpy(href.data(), d->ref, W * sizeof(long long),
                              hipMemcpyDeviceToHost));
    }
    if (confs) {
        memset(confs, 0, W * 2 * n * sizeof(double));
        if (d->stepped) {
            // yielded walkers are the parents selected by the cloning table;
            // the parent buffer of the last step is the one that is not `cur`
            const int par = 1 - d->cur;
            double *tmp;
            if (dev_alloc(&tmp, (size_t)nw * 2 * n)) return 1;
            long long tot = nw * (long long)n;
            hipLaunchKernelGGL(dmc_gather_state_kernel,
                               dim3((unsigned)((tot + 255) / 256)), dim3(256),
                               0, e->stream, d->pos[par], d->drift[par],
                               d->label[par], d->ref, nw, (int)n, tmp);
            HIP_TRY(hipMemcpyAsync(confs, tmp, (size_t)nw * 2 * n * 8,
                                   hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(hipStreamSynchronize(e->stream));
            hipFree(tmp);
        } else {
            std::vector<double> hp((size_t)nw * n), hd((size_t)nw * n);
            std::vector<unsigned short> hl((size_t)nw * n);
            HIP_TRY(hipMemcpy(hp.data(), d->pos[d->cur], hp.size() * 8,
                              hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(hd.data(), d->drift[d->cur], hd.size() * 8,
                              hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(hl.data(), d->label[d->cur], hl.size() * 2,
                              hipMemcpyDeviceToHost));
            for (size_t s = 0; s < (size_t)nw; ++s)
                for (size_t i = 0; i < n; ++i) {
                    confs[(s * 2 + 0) * n + hl[s * n + i]] = hp[s * n + i];
                    confs[(s * 2 + 1) * n + hl[s * n + i]] = hd[s * n + i];
                }
        }
    }
    if (energy) {
        memset(energy, 0, W * sizeof(double));
        // after a step eslot[s] holds the parent's energy (actual.energy)
        HIP_TRY(hipMemcpy(energy, d->stepped ? d->eslot : d->energy[d->cur],
                          (size_t)nw * 8, hipMemcpyDeviceToHost));
    }
    if (weight) {
        for (size_t s = 0; s < W; ++s) weight[s] = s < (size_t)nw ? 1.0 : 0.0;
        if (!d->stepped && d->init_weight.size() == (size_t)nw) {
            std::copy(d->init_weight.begin(), d->init_weight.end(), weight);
        } else if (!d->stepped) {
            HIP_TRY(hipMemcpy(weight, d->weight[d->cur], (size_t)nw * 8,
                              hipMemcpyDeviceToHost));
            for (size_t s = 0; s < (size_t)nw; ++s) weight[s] = std::exp(weight[s]);
        }
    }
    if (mask)
        for (size_t s = 0; s < W; ++s) mask[s] = s < (size_t)nw ? 0 : 1;
    if (cloning_ref)
        for (size_t s = 0; s < W; ++s) cloning_ref[s] = (int64_t)href[s];
    if (scalars) {
        scalars[0] = c.e_t; scalars[1] = c.w_t; scalars[2] = c.ref_energy;
        scalars[3] = c.total_weight != 0.0 ? c.total_energy / c.total_weight
                                           : 0.0;
        scalars[4] = (double)nw;
    }
    return 0;
}

static WalkerRecArgs walker_rec_args(qmc_dmc *d, long long first,
                                     long long count)
{
    WalkerRecArgs a;
    a.pos = d->pos[d->cur]; a.drift = d->drift[d->cur];
    a.label = d->label[d->cur];
    a.energy = d->energy[d->cur]; a.weight = d->weight[d->cur];
    a.eslot = d->eslot;
    // the rows the next estimator step reads as "previous"
    a.ssf_aux = d->est.num_modes > 0 && d->have_est
                    ? d->ssf_aux[d->est_last_act] : nullptr;
    a.dens_aux = d->est.num_bins > 0 && d->have_est
                     ? d->dens_aux[d->est_last_act] : nullptr;
    a.first = first; a.count = count;
    a.n = d->eng->dm.n;
    a.m3 = a.ssf_aux ? d->est.num_modes * 3 : 0;
    a.nb = a.dens_aux ? d->est.num_bins : 0;
    return a;
}

extern "C" int qmc_dmc_walker_record_size(qmc_dmc *d, int64_t *doubles)
{
    if (!d || !doubles) return fail("qmc_dmc_walker_record_size: null argument");
    const WalkerRecArgs a = walker_rec_args(d, 0, 0);
    *doubles = 3 * a.n + 2 + a.m3 + a.nb;
    return 0;
}

extern "C" int qmc_dmc_export_walkers(qmc_dmc *d, int64_t first, int64_t count,
                                      double *buf_dev)
{
    if (!d || !buf_dev) return fail("qmc_dmc_export_walkers: null argument");
    if (count <= 0) return 0;
    if (first < 0 || first + count > d->maxw)
        return fail("qmc_dmc_export_walkers: slot range outside the population");
    qmc_engine *e = d->eng;
    HIP_TRY(hipSetDevice(e->device));
    const WalkerRecArgs a = walker_rec_args(d, first, count);
    long long tot = count * (long long)(3 * a.n + 2 + a.m3 + a.nb);
    hipLaunchKernelGGL(pack_walkers_kernel, dim3((unsigned)((tot + 255) / 256)),
                       dim3(256), 0, e->stream, a, buf_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

// Stream-ordered: nothing here reads the device back.  The caller knows the
// population size (it gathered the counts to plan the rebalance).
extern "C" int qmc_dmc_import_walkers_at(qmc_dmc *d, int64_t first,
                                         int64_t count, const double *buf_dev)
{
    if (!d || !buf_dev) return fail("qmc_dmc_import_walkers_at: null argument");
    if (count <= 0) return 0;
    if (first < 0 || first + count > d->maxw)
        return fail("qmc_dmc_import_walkers_at: population would exceed "
                    "max_num_walkers");
    qmc_engine *e = d->eng;
    HIP_TRY(hipSetDevice(e->device));
    d->init_weight.clear();
    const WalkerRecArgs a = walker_rec_args(d, first, count);
    long long tot = count * (long long)(3 * a.n + 2 + a.m3 + a.nb);
    hipLaunchKernelGGL(unpack_walkers_kernel,
                       dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                       e->stream, a, buf_dev);
    // imported slots must not consume a spare normal stored for another walker
    hipLaunchKernelGGL(dmc_set_nw_kernel, dim3(1), dim3(64), 0, e->stream,
                       d->ctl, (long long)(first + count), (long long)first);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int qmc_dmc_set_num_walkers(qmc_dmc *d, int64_t nw)
{
    if (!d) return fail("qmc_dmc_set_num_walkers: null argument");
    if (nw < 0 || nw > d->maxw)
        return fail("qmc_dmc_set_num_walkers: bad population size");
    d->init_weight.clear();
    qmc_engine *e = d->eng;
    HIP_TRY(hipSetDevice(e->device));
    hipLaunchKernelGGL(dmc_set_nw_kernel, dim3(1), dim3(64), 0, e->stream,
                       d->ctl, (long long)nw, (long long)nw);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int qmc_dmc_import_walkers(qmc_dmc *d, int64_t count,
                                      const double *buf_dev)
{
    if (!d || !buf_dev) return fail("qmc_dmc_import_walkers: null argument");
    if (count <= 0) return 0;
    int64_t nw = 0;
    int rc = qmc_dmc_num_walkers(d, &nw);       // synchronises
    if (rc) return rc;
    if (nw + count > d->maxw)
        return fail("qmc_dmc_import_walkers: population would exceed "
                    "max_num_walkers");
    return qmc_dmc_import_walkers_at(d, nw, count, buf_dev);
}

extern "C" int qmc_dmc_truncate(qmc_dmc *d, int64_t new_nw)
{
    if (!d) return fail("qmc_dmc_truncate: null argument");
    HIP_TRY(hipSetDevice(d->eng->device));
    int64_t nw = 0;
    int rc = qmc_dmc_num_walkers(d, &nw);       // synchronises
    if (rc) return rc;
    if (new_nw < 0 || new_nw > nw)
        return fail("qmc_dmc_truncate: bad population size");
    return qmc_dmc_set_num_walkers(d, new_nw);
}
