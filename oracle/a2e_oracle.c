/*
 * a2e_oracle.c -- CPU restatement of SOC's stochastic-heating solver.  TEST INFRASTRUCTURE ONLY
 * (same rules as soc_oracle.c: only tests/, smoke() and bench legs may load it).
 *
 *   DoSolve ........ kernel_A2E.c:2-104   (transition matrix from absorptions x integration
 *                    weights, suffix sums, forward substitution with rescaling, emission)
 *   EqTemperature .. kernel_A2E.c:110-154 (trapezoid E_in, log-table T lookup, Planck emission)
 * -D NE/NFREQ/CELLS/NIP/FACTOR of A2E.py:283-286 are run-time arguments here.  Operand order
 * and float/double promotions are the reference's; compiled with -ffp-contract=off.
 * DoSolve uses +,*,/,max only, so both math modes give identical results for it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#ifdef SOC_ORACLE_LIBM
#  define M_EXP(x)     expf(x)
#  define M_LOG10(x)   log10f(x)
#  define M_POWN(x, n) powf((x), (float)(n))
#  define M_FLOOR(x)   floorf(x)
#else
#  include "../soc_amd/csrc/soc_math.h"
#  define M_EXP(x)     soc_expf(x)
#  define M_LOG10(x)   soc_log10f(x)
#  define M_POWN(x, n) soc_pownf((x), (n))
#  define M_FLOOR(x)   soc_floorf(x)
#endif

#define EXPORT __attribute__((visibility("default")))
#define IND(a, b) (((a) * (a) - (a)) / 2 + (b))

EXPORT int orc_a2e_dosolve(int batch, int NE, int NFREQ, const float *Iw, const int *L1, const int *L2,
                           const float *Tdown, const float *EA, const int *Ibeg, const float *AF,
                           const float *AABS, float *AEMIT)
{
    float *L = (float *)malloc(sizeof(float) * (size_t)((NE * NE - NE) / 2 + 1));
    float *XL = (float *)malloc(sizeof(float) * (size_t)NE);
    if (!L || !XL) return -1;
    for (int id = 0; id < batch; id++) {
        const float *ABS = AABS + (size_t)id * NFREQ;
        float *EMIT = AEMIT + (size_t)id * NFREQ;
        float I;
        int   iw_index = 0, j, u, l;
        for (l = 0; l < NE - 1; l++) {
            for (u = l + 1; u < NE; u++) {
                I = 0.0f;
                for (int i = L1[l * NE + u]; i <= L2[l * NE + u]; i++) {
                    I += ABS[i] * Iw[iw_index] * AF[i];
                    iw_index++;
                }
                L[IND(u, l)] = fmaxf(I, 0.0f);
            }
        }
        for (j = NE - 3; j > 0; j--) {
            u = j + 1;
            for (int i = 0; i < j; i++) L[IND(j, i)] += L[IND(u, i)];
        }
        XL[0] = 1.0e-20f;
        for (j = 1; j < NE; j++) {
            XL[j] = 0.0f;
            for (int i = 0; i <= j - 1; i++) XL[j] += L[IND(j, i)] * XL[i];
            XL[j] /= (Tdown[j] + 1.0e-30f);
            XL[j] = fmaxf(XL[j], 0.0f);
            if (XL[j] > 1.0e20f) {
                for (int i = 0; i <= j; i++) XL[i] *= 1.0e-20f;
            }
        }
        I = 0.0;
        for (int i = 0; i < NE; i++) I += XL[i];
        I = 1.0f / I;
        for (int i = 0; i < NE; i++) XL[i] = XL[i] * I;
        for (j = 0; j < NFREQ; j++) {
            I = 0.0f;
            for (int i = Ibeg[j]; i < NE; i++) I += EA[j * NE + i] * XL[i];
            EMIT[j] = I;
        }
    }
    free(L);
    free(XL);
    return 0;
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

EXPORT void orc_a2e_eqtemp(int batch, int icell, int CELLS, int NFREQ, int NIP, float FACTOR, float kE, float oplgkE,
                           float Emin, const float *FREQ, const float *KABS, const float *TTT, const float *ABS,
                           float *T, float *EMIT)
{
    for (int id = 0; id < batch; id++) {
        int ind = icell + id, iE;
        if (ind >= CELLS) continue;
        float Ein, wi, TP, f;
        const float *A = ABS + (size_t)id * NFREQ;
        Ein = 0.0f;
        for (int i = 1; i < NFREQ; i++)
            Ein += (A[i] * FREQ[i] + A[i - 1] * FREQ[i - 1]) * ((FREQ[i] - FREQ[i - 1]) * 3.3130348e-27f);
        iE = clampi((int)M_FLOOR(oplgkE * M_LOG10(Ein / Emin)), 0, NIP - 2);
        wi = (Emin * M_POWN(kE, iE + 1) - Ein) / (Emin * M_POWN(kE, iE + 1) - M_POWN(kE, iE));
        TP = wi * TTT[iE] + (1.0 - wi) * TTT[iE + 1];
        T[id] = TP;
        for (int ifreq = 0; ifreq < NFREQ; ifreq++) {
            f = FREQ[ifreq];
            EMIT[(size_t)id * NFREQ + ifreq] =
                (2.79639459e-20f * FACTOR) * KABS[ifreq] * (f * f / (M_EXP(4.7995074e-11f * f / TP) - 1.0f));
        }
    }
}

/* kernel_eqsolver.c (A2E_MABU.py SolveEquilibriumDust, :436-640): equilibrium temperature of one dust component
 * from its share of the absorptions, EqTemperature :5-62 -- Ein by trapezoid with the end intervals counted once,
 * lookup with 0.5*Ein (beta = 1), T = 2.7 for cells without absorbed energy -- and Emission :66-79 for every
 * frequency (non-finite values -> 0).  -D CELLS/NFREQ/FACTOR (A2E_MABU.py:492) are arguments; CR_HEATING = 0. */
EXPORT void orc_eqsolver(int batch, int icell, int CELLS, int NFREQ, int NE, float FACTOR, float kE, float oplgkE, float Emin,
                         const float *FREQ, const float *KABS, const float *TTT, const float *ABS, float *T, float *EMIT)
{
    for (int id = 0; id < batch; id++) {
        int ind = icell + id, iE;
        if (ind >= CELLS) continue;
        const float scale = 6.62607e-27f;
        float wi, beta = 1.0f, Ein = 0.0f, TP;
        const float *A = ABS + (size_t)id * NFREQ;
        Ein += A[0] * FREQ[0] * scale * (FREQ[1] - FREQ[0]);
        Ein += A[NFREQ - 1] * FREQ[NFREQ - 1] * scale * (FREQ[NFREQ - 1] - FREQ[NFREQ - 2]);
        for (int i = 1; i < (NFREQ - 1); i++) Ein += A[i] * FREQ[i] * scale * (FREQ[i + 1] - FREQ[i - 1]);
        iE = clampi((int)M_FLOOR(oplgkE * M_LOG10((0.5f * Ein / beta) / Emin)), 0, NE - 2);
        wi = (Emin * M_POWN(kE, iE + 1) - (Ein / beta)) / (Emin * M_POWN(kE, iE + 1) - M_POWN(kE, iE));
        TP = wi * TTT[iE] + (1.0 - wi) * TTT[iE + 1];
        if (Ein <= 0.0f) TP = 2.7f;
        T[id] = TP;
        for (int f = 0; f < NFREQ; f++) {
            const float res = (2.79639459e-20f * FACTOR) * KABS[f] * (FREQ[f] * FREQ[f] / (M_EXP(4.7995074e-11f * FREQ[f] / TP) - 1.0f));
            EMIT[(size_t)id * NFREQ + f] = isfinite(res) ? res : 0.0f;
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * The two kernels of kernel_A2E_pre.c that A2E_pre.py runs per grain size when it writes a solver file.
 * FACTOR (-D, A2E_pre.py:134) is an argument.  PLANCK, BOLTZMANN are the kernel file's float literals (:1-3).
 * ------------------------------------------------------------------------------------------------ */
#define PRE_BOLTZMANN (1.3806488e-16f)
#define PRE_PLANCK    (6.6260696e-27f)
#define PRE_SS        8

/* kernel_A2E_pre.c:10-23: linear interpolation, no extrapolation; float arithmetic */
static float pre_interpolate(const int n, const float *x, const float *y, const float x0)
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

static inline double pre_clampd(double x, double lo, double hi) { const double m = (x < lo) ? lo : x;  return (hi < m) ? hi : m; }
static inline double pre_mind(double a, double b) { return (b < a) ? b : a; }
static inline double pre_maxd(double a, double b) { return (a < b) ? b : a; }

/* PrepareTdown (kernel_A2E_pre.c:123-212): cooling rates u -> u-1 in the thermal continuous approximation; one u per
 * work item, u = 1 .. NE-1; Tdown[0] = 0.  The mixed float / double arithmetic is the kernel's. */
EXPORT void orc_a2e_pre_tdown(int NFREQ, const float *FREQ, const float *Ef, const float *SKABS, int NE, const float *E,
                              const float *T, float *Tdown)
{
    Tdown[0] = 0.0f;
    for (int u = 1; u < NE; u++) {
        double Tu, I, ee0, ee1, yy0, yy1, Eu, El, x;
        int i;
        Eu  = 0.5 * (E[u] + E[u + 1]);             /* float sum, double product: "0.5*(E[u]+E[u+1])" */
        El  = 0.5 * (E[u - 1] + E[u]);
        Tu  = pre_interpolate(NE + 1, E, T, (float)Eu);
        ee0 = 0.0;
        yy0 = 0.0;
        i   = 0;
        I   = 0.0;
        while ((i < (NFREQ - 1)) && Ef[i + 1] < Eu) {
            ee0 = Ef[i];
            x   = pre_interpolate(NFREQ, FREQ, SKABS, (float)(ee0 / PRE_PLANCK));
            yy0 = ee0 * ee0 * ee0 * x / (exp(ee0 / (PRE_BOLTZMANN * Tu)) - 1.0);
            for (int j = 0; j < PRE_SS; j++) {
                ee1 = Ef[i] + (j + 1) * (Ef[i + 1] - Ef[i]) / PRE_SS;     /* float arithmetic, then double */
                x   = pre_interpolate(NFREQ, FREQ, SKABS, (float)(ee1 / PRE_PLANCK));
                yy1 = ee1 * ee1 * ee1 * x / (exp(ee1 / (PRE_BOLTZMANN * Tu)) - 1.0);
                I  += 0.5 * (ee1 - ee0) * (yy1 + yy0);
                ee0 = ee1;
                yy0 = yy1;
            }
            i++;
        }
        if (i < (NFREQ - 1)) {
            for (int j = 0; j < PRE_SS; j++) {
                ee1 = Ef[i] + (j + 1) * (Eu - Ef[i]) / PRE_SS;             /* double: Eu is double */
                x   = pre_interpolate(NFREQ, FREQ, SKABS, (float)(ee1 / PRE_PLANCK));
                yy1 = ee1 * ee1 * ee1 * x / (exp(ee1 / (PRE_BOLTZMANN * Tu)) - 1.0);
                I  += 0.5 * (ee1 - ee0) * (yy1 + yy0);
                ee0 = ee1;
                yy0 = yy1;
            }
        }
        I *= 9.612370e+58 / (Eu - El);
        Tdown[u] = (float)I;
    }
}

/* PrepareIntegrationWeightsTrapezoid (kernel_A2E_pre.c:580-736): for every pair of enthalpy bins l < u the weights
 * with which the absorbed photons of the frequency grid feed the transition l -> u (trapezoid rule over the four
 * breakpoints W1..W4 of the bin overlap, plus the intrabin term for u = l+1).  One l per work item.
 * IW: NE*NFREQ floats per l (the weights of its pairs, one after the other), noIw[l] how many; L1, L2[l*NE+u] first and
 * last frequency of the pair (-1, -2: none).  wrk: NFREQ floats per l. */
EXPORT void orc_a2e_pre_weights(int NFREQ, int NE, float FACTOR, const float *Ef, const float *E, int *L1, int *L2,
                                float *IW, float *wrk, int *noIw)
{
    for (int l = 0; l < NE - 1; l++) {
        int index = 0;
        double El, Eu, dEl, dEu, coeff, alpha, beta, G1, G2;
        double W1, W2, W3, W4, a, b;
        int i;
        float *temp_Iw = &wrk[(size_t)l * NFREQ];
        float *Iw = &IW[(size_t)l * NE * NFREQ];
        El  = 0.5 * (E[l] + E[l + 1]);
        dEl = E[l + 1] - E[l];
        for (int u = l + 1; u < NE; u++) {
            Eu  = 0.5 * (E[u] + E[u + 1]);
            dEu = E[u + 1] - E[u];
            W1  = E[u] - E[l + 1];
            W2  = fminf(E[u] - E[l], E[u + 1] - E[l + 1]);          /* min / max of floats (:622-623) */
            W3  = fmaxf(E[u] - E[l], E[u + 1] - E[l + 1]);
            W4  = E[u + 1] - E[l];
            if ((Ef[0] > W4) || (Ef[NFREQ - 1] < W1)) {
                L1[l * NE + u] = -1;
                L2[l * NE + u] = -2;
                continue;
            }
            for (i = 0; i < NFREQ; i++) temp_Iw[i] = 0.0f;
            coeff = 1.0 / (Eu - El) / (FACTOR * PRE_PLANCK);          /* float product FACTOR*PLANCK */
            i = 1;
            while ((i < (NFREQ - 1)) && (Ef[i] < W1)) i += 1;
            i = (i - 1 > 0) ? (i - 1) : 0;
            /* W1 - W2 */
            a     = pre_clampd(W1, (double)Ef[i], (double)Ef[i + 1]);
            b     = pre_clampd(W2, a, (double)Ef[i + 1]);
            alpha = (a - Ef[i]) / (Ef[i + 1] - Ef[i]);
            beta  = (b - Ef[i]) / (Ef[i + 1] - Ef[i]);
            G1    = (a - W1) / dEl;
            G2    = (b - W1) / dEl;
            temp_Iw[i]     += 0.5 * (b - a) * (G1 * a * (1.0 - alpha) + G2 * b * (1.0 - beta)) * coeff;
            temp_Iw[i + 1] += 0.5 * (b - a) * (G1 * a * alpha + G2 * b * beta) * coeff;
            if (b < W2) i += 1;
            while ((i < (NFREQ - 1)) && (b < W2)) {
                a     = b;
                G1    = G2;
                b     = pre_mind(W2, (double)Ef[i + 1]);
                alpha = (a - Ef[i]) / (Ef[i + 1] - Ef[i]);
                beta  = (b - Ef[i]) / (Ef[i + 1] - Ef[i]);
                G2    = (b - W1) / dEl;
                temp_Iw[i]     += 0.5 * (b - a) * (G1 * a * (1.0 - alpha) + G2 * b * (1.0 - beta)) * coeff;
                temp_Iw[i + 1] += 0.5 * (b - a) * (G1 * a * alpha + G2 * b * beta) * coeff;
                if (b < W2) i += 1;
            }
            /* W2 - W3 */
            while ((i < (NFREQ - 1)) && (b < W3)) {
                a     = b;
                G1    = G2;
                b     = pre_mind(W3, (double)Ef[i + 1]);
                G2    = pre_mind(dEl, dEu) / dEl;
                alpha = (a - Ef[i]) / (Ef[i + 1] - Ef[i]);
                beta  = (b - Ef[i]) / (Ef[i + 1] - Ef[i]);
                temp_Iw[i]     += 0.5 * (b - a) * (G1 * a * (1.0 - alpha) + G2 * b * (1.0 - beta)) * coeff;
                temp_Iw[i + 1] += 0.5 * (b - a) * (G1 * a * alpha + G2 * b * beta) * coeff;
                if (b < W3) i += 1;
            }
            /* W3 - W4 */
            while ((i < (NFREQ - 1)) && (b < W4)) {
                a     = b;
                G1    = G2;
                b     = pre_mind(W4, (double)Ef[i + 1]);
                alpha = (a - Ef[i]) / (Ef[i + 1] - Ef[i]);
                beta  = (b - Ef[i]) / (Ef[i + 1] - Ef[i]);
                G2    = (W4 - 0.5 * (a + b)) / dEl;
                temp_Iw[i]     += 0.5 * (b - a) * (G1 * a * (1.0 - alpha) + G2 * b * (1.0 - beta)) * coeff;
                temp_Iw[i + 1] += 0.5 * (b - a) * (G1 * a * alpha + G2 * b * beta) * coeff;
                if (b < W4) i += 1;
            }
            /* intrabin (:699-715) */
            if (u == (l + 1)) {
                i = 0;
                b = Ef[0];
                while ((i < (NFREQ - 1)) && (Ef[i] < dEl)) {
                    a     = b;
                    b     = pre_clampd(dEl, a, (double)Ef[i + 1]);
                    alpha = (a - Ef[i]) / (Ef[i + 1] - Ef[i]);
                    beta  = (b - Ef[i]) / (Ef[i + 1] - Ef[i]);
                    temp_Iw[i]     += 0.5 * (b - a) * ((1.0 - a / dEl) * a * (1.0 - alpha) + (1.0 - b / dEl) * b * (1.0 - beta)) * coeff;
                    temp_Iw[i + 1] += 0.5 * (b - a) * ((1.0 - a / dEl) * a * alpha + (1.0 - b / dEl) * b * beta) * coeff;
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
            for (i = L1[l * NE + u]; i <= L2[l * NE + u]; i++) {
                if (i < NFREQ) Iw[index] = temp_Iw[i];
                else           Iw[index] = 0.0f;
                index++;
            }
        }
        noIw[l] = index;
    }
}
