// soc_a2e_pre.hip -- what A2E_pre.py computes per grain size when it writes a <dust>.solver file: the integration
// weights of the transitions between enthalpy bins (PrepareIntegrationWeightsTrapezoid, kernel_A2E_pre.c:580-736) and the
// cooling rates of the thermal continuous approximation (PrepareTdown, :123-212).
//
// One-off preprocessing (seconds per dust model): the launch shape is the reference's -- one lane per lower bin l
// (weights), one per upper bin u (cooling) -- and the arithmetic its mix of float and double, operation for operation,
// so that a solver file written here equals one written by A2E_pre.py (weights, L1, L2: bit for bit; the cooling rates
// go through exp() in double, which the device evaluates with its own library).
#include "soc_dev.h"

#define PRE_BOLTZMANN (1.3806488e-16f)
#define PRE_PLANCK    (6.6260696e-27f)
#define PRE_SS        8

// Interpolate (kernel_A2E_pre.c:10-23)
__device__ __forceinline__ float soc_pre_interpolate(const int n, const float *x, const float *y, const float x0)
{
    if (x0 <= x[0])     return y[0];
    if (x0 >= x[n - 1]) return y[n - 1];
    int a = 0, c = n - 1, b;
    while ((c - a) > 4) {
        b = (a + c) / 2;
        if (x[b] > x0) c = b; else a = b;
    }
    for (b = a; b <= c; b++) if (x[b] >= x0) break;
    const float w = (x[b] - x0) / (x[b] - x[b - 1]);
    return w * y[b - 1] + (1.0f - w) * y[b];
}

__device__ __forceinline__ double soc_pre_clamp(double x, double lo, double hi) { const double m = (x < lo) ? lo : x;  return (hi < m) ? hi : m; }
__device__ __forceinline__ double soc_pre_min(double a, double b) { return (b < a) ? b : a; }

__global__ __launch_bounds__(64) void soc_a2e_pre_tdown_kernel(const int NFREQ, const float *FREQ, const float *Ef, const float *SKABS,
                                                               const int NE, const float *E, const float *T, float *Tdown)
{
    const int u = 1 + (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (u >= NE) return;
    if (u == 1) Tdown[0] = 0.0f;
    double I = 0.0, ee0 = 0.0, ee1, yy0 = 0.0, yy1, x;
    const double Eu = 0.5 * (E[u] + E[u + 1]);
    const double El = 0.5 * (E[u - 1] + E[u]);
    const double Tu = soc_pre_interpolate(NE + 1, E, T, (float)Eu);
    int i = 0;
    while ((i < (NFREQ - 1)) && Ef[i + 1] < Eu) {
        ee0 = Ef[i];
        x   = soc_pre_interpolate(NFREQ, FREQ, SKABS, (float)(ee0 / PRE_PLANCK));
        yy0 = ee0 * ee0 * ee0 * x / (exp(ee0 / (PRE_BOLTZMANN * Tu)) - 1.0);
        for (int j = 0; j < PRE_SS; j++) {
            ee1 = Ef[i] + (j + 1) * (Ef[i + 1] - Ef[i]) / PRE_SS;           // float arithmetic, as written
            x   = soc_pre_interpolate(NFREQ, FREQ, SKABS, (float)(ee1 / PRE_PLANCK));
            yy1 = ee1 * ee1 * ee1 * x / (exp(ee1 / (PRE_BOLTZMANN * Tu)) - 1.0);
            I  += 0.5 * (ee1 - ee0) * (yy1 + yy0);
            ee0 = ee1;
            yy0 = yy1;
        }
        i++;
    }
    if (i < (NFREQ - 1)) {
        for (int j = 0; j < PRE_SS; j++) {
            ee1 = Ef[i] + (j + 1) * (Eu - Ef[i]) / PRE_SS;
            x   = soc_pre_interpolate(NFREQ, FREQ, SKABS, (float)(ee1 / PRE_PLANCK));
            yy1 = ee1 * ee1 * ee1 * x / (exp(ee1 / (PRE_BOLTZMANN * Tu)) - 1.0);
            I  += 0.5 * (ee1 - ee0) * (yy1 + yy0);
            ee0 = ee1;
            yy0 = yy1;
        }
    }
    I *= 9.612370e+58 / (Eu - El);
    Tdown[u] = (float)I;
}

// the contribution of one piece [a, b] of a frequency bin to the two weights at its ends (the four statements the
// reference repeats in every section of the integral)
#define SOC_PRE_ADD(F1, F2) do { \
        temp_Iw[i]     += 0.5 * (b - a) * ((F1) * a * (1.0 - alpha) + (F2) * b * (1.0 - beta)) * coeff; \
        temp_Iw[i + 1] += 0.5 * (b - a) * ((F1) * a * alpha + (F2) * b * beta) * coeff; } while (0)

__global__ __launch_bounds__(64) void soc_a2e_pre_weights_kernel(const int NFREQ, const int NE, const float FACTOR, const float *Ef, const float *E,
                                                                 int *L1, int *L2, float *IW, float *wrk, int *noIw)
{
    const int l = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (l >= (NE - 1)) return;
    int index = 0, i;
    float *temp_Iw = &wrk[(size_t)l * NFREQ];
    float *Iw = &IW[(size_t)l * NE * NFREQ];
    const double El = 0.5 * (E[l] + E[l + 1]);
    const double dEl = E[l + 1] - E[l];
    for (int u = l + 1; u < NE; u++) {
        const double Eu = 0.5 * (E[u] + E[u + 1]);
        const double dEu = E[u + 1] - E[u];
        const double W1 = E[u] - E[l + 1];
        const double W2 = fminf(E[u] - E[l], E[u + 1] - E[l + 1]);
        const double W3 = fmaxf(E[u] - E[l], E[u + 1] - E[l + 1]);
        const double W4 = E[u + 1] - E[l];
        if ((Ef[0] > W4) || (Ef[NFREQ - 1] < W1)) {
            L1[l * NE + u] = -1;
            L2[l * NE + u] = -2;
            continue;
        }
        for (i = 0; i < NFREQ; i++) temp_Iw[i] = 0.0f;
        const double coeff = 1.0 / (Eu - El) / (FACTOR * PRE_PLANCK);
        double a, b, alpha, beta, G1, G2;
        i = 1;
        while ((i < (NFREQ - 1)) && (Ef[i] < W1)) i += 1;
        i = (i - 1 > 0) ? (i - 1) : 0;
        // The integrand G(E) E C_abs(E) over [W1, W4] is piecewise: G rises on [W1, W2], is flat on [W2, W3], falls on
        // [W3, W4].  Every section is cut at the frequency grid; a piece [a, b] adds to the weights of its two grid
        // points.  The first piece starts inside the bin of W1, the others continue from where the last one ended.
        a     = soc_pre_clamp(W1, (double)Ef[i], (double)Ef[i + 1]);
        b     = soc_pre_clamp(W2, a, (double)Ef[i + 1]);
        alpha = (a - Ef[i]) / (Ef[i + 1] - Ef[i]);
        beta  = (b - Ef[i]) / (Ef[i + 1] - Ef[i]);
        G1    = (a - W1) / dEl;
        G2    = (b - W1) / dEl;
        SOC_PRE_ADD(G1, G2);
        if (b < W2) i += 1;
        for (int section = 0; section < 3; section++) {
            const double Wend = (section == 0) ? W2 : ((section == 1) ? W3 : W4);
            while ((i < (NFREQ - 1)) && (b < Wend)) {
                a     = b;
                G1    = G2;
                b     = soc_pre_min(Wend, (double)Ef[i + 1]);
                alpha = (a - Ef[i]) / (Ef[i + 1] - Ef[i]);
                beta  = (b - Ef[i]) / (Ef[i + 1] - Ef[i]);
                G2    = (section == 0) ? ((b - W1) / dEl) : ((section == 1) ? (soc_pre_min(dEl, dEu) / dEl) : ((W4 - 0.5 * (a + b)) / dEl));
                SOC_PRE_ADD(G1, G2);
                if (b < Wend) i += 1;
            }
        }
        // inside the bin (u = l + 1)
        if (u == (l + 1)) {
            i = 0;
            b = Ef[0];
            while ((i < (NFREQ - 1)) && (Ef[i] < dEl)) {
                a     = b;
                b     = soc_pre_clamp(dEl, a, (double)Ef[i + 1]);
                alpha = (a - Ef[i]) / (Ef[i + 1] - Ef[i]);
                beta  = (b - Ef[i]) / (Ef[i + 1] - Ef[i]);
                SOC_PRE_ADD(1.0 - a / dEl, 1.0 - b / dEl);
                i += 1;
            }
        }
        int first_non_zero = -1, last_non_zero = -2;
        for (i = 0; i < NFREQ; i++) {
            if (temp_Iw[i] > 0.0 && first_non_zero < 0) first_non_zero = i;
            if (temp_Iw[i] > 0.0) last_non_zero = i;
        }
        L1[l * NE + u] = first_non_zero;
        L2[l * NE + u] = last_non_zero;
        for (i = first_non_zero; i <= last_non_zero; i++) {
            Iw[index] = (i < NFREQ) ? temp_Iw[i] : 0.0f;
            index++;
        }
    }
    noIw[l] = index;
}

hipError_t soc_launch_a2e_pre(int NFREQ, int NE, float FACTOR, const float *FREQ, const float *Ef, const float *SKABS, const float *E, const float *T,
                              int *L1, int *L2, float *IW, float *wrk, int *noIw, float *Tdown, hipStream_t st)
{
    const int nb = (NE + 63) / 64;
    soc_a2e_pre_weights_kernel<<<nb, 64, 0, st>>>(NFREQ, NE, FACTOR, Ef, E, L1, L2, IW, wrk, noIw);
    soc_a2e_pre_tdown_kernel<<<nb, 64, 0, st>>>(NFREQ, FREQ, Ef, SKABS, NE, E, T, Tdown);
    return hipGetLastError();
}
