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
