// soc_a2e_pre.hip -- what A2E_pre.py computes per grain size when it writes a <dust>.solver file: the integration
// weights of the transitions between enthalpy bins (the quantities of PrepareIntegrationWeightsTrapezoid,
// kernel_A2E_pre.c:580-736) and the cooling rates of the thermal continuous approximation (PrepareTdown, :123-212).
//
// The reference gives one work item a lower bin l and lets it walk all upper bins u one after the other, the packed
// weights of l growing behind a running index.  The transitions (l, u) are independent, so here
//   * soc_pre_weights_kernel: one WORKGROUP per lower bin, one LANE per upper bin.  A lane integrates its transition's
//     window function over the frequency grid into a column of LDS (NFREQ floats per lane, lane-minor: no bank
//     conflicts), finds the first and the last frequency that got weight, and a workgroup-wide prefix sum over the
//     window lengths gives every transition its place in l's packed segment -- no serial index;
//   * soc_pre_cooling_kernel: one WAVE per upper bin, one lane per frequency interval (its eight sub-steps), the
//     interval sums added in double by a wave reduction.
// Arithmetic contract with the reference (tests/test_a2e_pre.py: weights, L1, L2 bit for bit against the x86 build of
// kernel_A2E_pre.c): every accumulation into a weight is `float += double`, and the pieces of a transition's window are
// deposited in ascending energy, as there.  The cooling rates go through exp() in double, which the device evaluates with
// its own library, and are summed in another order: equal to ~1e-15 relative, compared at 1e-6.
#include "soc_dev.h"

#define PRE_BOLTZMANN (1.3806488e-16f)
#define PRE_PLANCK    (6.6260696e-27f)
#define PRE_SUBSTEPS  8
#define PRE_T         64          // lanes of the weights workgroup = upper bins in flight

// y(x0) on the table (x, y), linear, clamped at the ends (kernel_A2E_pre.c:10-23 `Interpolate`: bisection down to a
// window of at most five nodes, then the first node at or above x0)
__device__ __forceinline__ float soc_pre_table(const int n, const float *x, const float *y, const float x0)
{
    if (x0 <= x[0])     return y[0];
    if (x0 >= x[n - 1]) return y[n - 1];
    int lo = 0, hi = n - 1;
    while (hi - lo > 4) {
        const int mid = (lo + hi) / 2;
        if (x[mid] > x0) hi = mid; else lo = mid;
    }
    int k = lo;
    while (k <= hi && !(x[k] >= x0)) k++;
    const float w = (x[k] - x0) / (x[k] - x[k - 1]);
    return w * y[k - 1] + (1.0f - w) * y[k];
}

__device__ __forceinline__ double soc_pre_within(double x, double lo, double hi) { const double m = (x < lo) ? lo : x;  return (hi < m) ? hi : m; }

// One transition l -> u.  The window function of the transition (how much of bin u a photon of energy x lifts a grain of
// bin l into) is a trapezoid over [W1, W4]: it rises from W1 to W2, is flat to W3, falls to W4 (:617-623).  It is integrated
// against x on the frequency grid by cutting [W1, W4] at the grid nodes and at W2, W3; a piece [x0, x1] inside the grid
// interval `node` deposits onto the two nodes of that interval with the hat-function weights of its end points.
struct SocPreWindow {
    float *col;                  // this lane's accumulators: col[f * PRE_T]
    const float *Ef;
    int    nfreq;
    double scale;                // 1 / (Eu - El) / (FACTOR * PLANCK)

    __device__ __forceinline__ void deposit(const int node, const double x0, const double x1, const double g0, const double g1)
    {
        const double width = Ef[node + 1] - Ef[node];
        const double h0 = (x0 - Ef[node]) / width, h1 = (x1 - Ef[node]) / width;      // hat weights of the upper node
        col[node * PRE_T]       += 0.5 * (x1 - x0) * (g0 * x0 * (1.0 - h0) + g1 * x1 * (1.0 - h1)) * scale;
        col[(node + 1) * PRE_T] += 0.5 * (x1 - x0) * (g0 * x0 * h0 + g1 * x1 * h1) * scale;
    }
};

__global__ __launch_bounds__(PRE_T) void soc_pre_weights_kernel(const int NFREQ, const int NE, const float FACTOR, const float *Ef, const float *E,
                                                                int *L1, int *L2, float *IW, int *noIw)
{
    extern __shared__ float lds[];
    float *acc  = lds;                                    // [NFREQ][PRE_T]
    int   *scan = (int *)(acc + (size_t)NFREQ * PRE_T);   // [PRE_T] window lengths -> places
    const int l = (int)blockIdx.x, lane = (int)threadIdx.x;
    if (l >= NE - 1) return;
    float *segment = IW + (size_t)l * NE * NFREQ;         // the packed weights of lower bin l
    const double El = 0.5 * (E[l] + E[l + 1]), dEl = E[l + 1] - E[l];
    int filled = 0;                                       // weights of l packed so far (the same in every lane)
    for (int u0 = l + 1; u0 < NE; u0 += PRE_T) {
        const int  u = u0 + lane;
        const bool on = (u < NE);
        int first = -1, last = -2;
        if (on) {
            const double Eu = 0.5 * (E[u] + E[u + 1]), dEu = E[u + 1] - E[u];
            const double W1 = E[u] - E[l + 1];
            const double W2 = fminf(E[u] - E[l], E[u + 1] - E[l + 1]);
            const double W3 = fmaxf(E[u] - E[l], E[u + 1] - E[l + 1]);
            const double W4 = E[u + 1] - E[l];
            if (!((Ef[0] > W4) || (Ef[NFREQ - 1] < W1))) {                  // the window meets the simulated frequencies
                SocPreWindow win;
                win.col = acc + lane;  win.Ef = Ef;  win.nfreq = NFREQ;
                win.scale = 1.0 / (Eu - El) / (FACTOR * PRE_PLANCK);
                for (int f = 0; f < NFREQ; f++) win.col[f * PRE_T] = 0.0f;
                // the grid interval that holds W1 (:631-636)
                int node = 1;
                while ((node < NFREQ - 1) && (Ef[node] < W1)) node++;
                node = (node > 1) ? (node - 1) : 0;
                // three stretches of the window, each ending at `stop`; `level` is the window function at the right end of a piece
                double x0, x1 = soc_pre_within(W1, (double)Ef[node], (double)Ef[node + 1]), g0, g1 = (x1 - W1) / dEl;
                const double plateau = ((dEu < dEl) ? dEu : dEl) / dEl;
                bool opened = false;                       // the first piece starts inside its interval; the later ones where the last ended
                for (int stretch = 0; stretch < 3; stretch++) {
                    const double stop = (stretch == 0) ? W2 : ((stretch == 1) ? W3 : W4);
                    while (true) {
                        if (opened && !((node < NFREQ - 1) && (x1 < stop))) break;
                        x0 = x1;  g0 = g1;
                        x1 = opened ? ((stop < (double)Ef[node + 1]) ? stop : (double)Ef[node + 1]) : soc_pre_within(stop, x0, (double)Ef[node + 1]);
                        g1 = (stretch == 0) ? ((x1 - W1) / dEl) : ((stretch == 1) ? plateau : ((W4 - 0.5 * (x0 + x1)) / dEl));
                        win.deposit(node, x0, x1, g0, g1);
                        opened = true;
                        if (x1 < stop) node++;
                    }
                }
                // absorptions that keep the grain inside its bin count for the step to the next one (u = l + 1, :704-719)
                if (u == l + 1) {
                    node = 0;
                    x1 = Ef[0];
                    while ((node < NFREQ - 1) && (Ef[node] < dEl)) {
                        x0 = x1;
                        x1 = soc_pre_within(dEl, x0, (double)Ef[node + 1]);
                        win.deposit(node, x0, x1, 1.0 - x0 / dEl, 1.0 - x1 / dEl);
                        node++;
                    }
                }
                for (int f = 0; f < NFREQ; f++)
                    if (win.col[f * PRE_T] > 0.0f) { if (first < 0) first = f;  last = f; }
            } else {
                first = -1;  last = -2;
            }
            L1[l * NE + u] = first;
            L2[l * NE + u] = last;
        }
        // places of the windows in l's segment: exclusive prefix sum of their lengths over the lanes
        const int len = on ? (last - first + 1) : 0;
        scan[lane] = len;
        __syncthreads();
        for (int d = 1; d < PRE_T; d <<= 1) {
            const int v = (lane >= d) ? scan[lane - d] : 0;
            __syncthreads();
            scan[lane] += v;
            __syncthreads();
        }
        const int place = filled + scan[lane] - len;
        const int total = scan[PRE_T - 1];
        for (int k = 0; k < len; k++) segment[place + k] = acc[(size_t)(first + k) * PRE_T + lane];
        filled += total;
        __syncthreads();
    }
    if (lane == 0) noIw[l] = filled;
}

// the integrand of the cooling rate at photon energy e for a grain at temperature Tu (:150-152)
__device__ __forceinline__ double soc_pre_emission(const int NFREQ, const float *FREQ, const float *SKABS, const double e, const double Tu)
{
    const double k = soc_pre_table(NFREQ, FREQ, SKABS, (float)(e / PRE_PLANCK));
    return e * e * e * k / (exp(e / (PRE_BOLTZMANN * Tu)) - 1.0);
}

__global__ __launch_bounds__(64) void soc_pre_cooling_kernel(const int NFREQ, const float *FREQ, const float *Ef, const float *SKABS,
                                                             const int NE, const float *E, const float *T, float *Tdown)
{
    const int u = 1 + (int)blockIdx.x, lane = (int)threadIdx.x;
    if (u >= NE) return;
    if (u == 1 && lane == 0) Tdown[0] = 0.0f;
    const double Eu = 0.5 * (E[u] + E[u + 1]), El = 0.5 * (E[u - 1] + E[u]);
    const double Tu = soc_pre_table(NE + 1, E, T, (float)Eu);
    // the frequency intervals below Eu are integrated whole, the one that holds Eu up to Eu (:146-181)
    int whole = 0;
    while ((whole < NFREQ - 1) && (Ef[whole + 1] < Eu)) whole++;
    double mine = 0.0;
    for (int i = lane; i <= whole && i < NFREQ - 1; i += 64) {
        const bool cut = (i == whole);                      // the interval of Eu
        double e0, y0;
        if (!cut) {
            e0 = Ef[i];
            y0 = soc_pre_emission(NFREQ, FREQ, SKABS, e0, Tu);
        } else if (i == 0) {
            e0 = 0.0;  y0 = 0.0;                            // (nothing before it: the reference starts this piece from zero)
        } else {
            e0 = Ef[i - 1] + PRE_SUBSTEPS * (Ef[i] - Ef[i - 1]) / PRE_SUBSTEPS;      // where the interval before ended, in float as there
            y0 = soc_pre_emission(NFREQ, FREQ, SKABS, e0, Tu);
        }
        for (int j = 0; j < PRE_SUBSTEPS; j++) {
            const double e1 = cut ? (Ef[i] + (j + 1) * (Eu - Ef[i]) / PRE_SUBSTEPS) : (double)(Ef[i] + (j + 1) * (Ef[i + 1] - Ef[i]) / PRE_SUBSTEPS);
            const double y1 = soc_pre_emission(NFREQ, FREQ, SKABS, e1, Tu);
            mine += 0.5 * (e1 - e0) * (y1 + y0);
            e0 = e1;  y0 = y1;
        }
    }
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if (lane == 0) Tdown[u] = (float)(mine * (9.612370e+58 / (Eu - El)));
}

hipError_t soc_launch_a2e_pre(int NFREQ, int NE, float FACTOR, const float *FREQ, const float *Ef, const float *SKABS, const float *E, const float *T,
                              int *L1, int *L2, float *IW, float *wrk, int *noIw, float *Tdown, hipStream_t st)
{
    (void)wrk;                                             // (the accumulators live in LDS)
    const size_t lds = ((size_t)NFREQ * PRE_T + PRE_T) * 4;
    if (lds > 160 * 1024) return hipErrorInvalidValue;     // NFREQ <= 639
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)soc_pre_weights_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    soc_pre_weights_kernel<<<NE - 1, PRE_T, lds, st>>>(NFREQ, NE, FACTOR, Ef, E, L1, L2, IW, noIw);
    soc_pre_cooling_kernel<<<NE - 1, 64, 0, st>>>(NFREQ, FREQ, Ef, SKABS, NE, E, T, Tdown);
    return hipGetLastError();
}
