// soc_a2e.hip -- stochastically heated grains: DoSolve and EqTemperature of kernel_A2E.c for gfx950.
//
// Reference shape (kernel_A2E.c:2-104): one work item per cell, the lower-triangular
// transition matrix L[(NE^2-NE)/2] of every cell spilled to a global scratch buffer
// interleaved by LOCAL (NE=128 -> 266 MB per batch of 8192 cells; the source comments put
// most of the run time into building and re-reading it), a private XL[NE].
//
// Here: ONE WAVE PER CELL, L lives in LDS (32.5 KB at NE=128) and never touches HBM:
//   1. heating rates: the (l,u) pairs are spread over the 64 lanes; each entry is the
//      reference's sequential sum over its frequency window (host-built pair tables give the
//      window, the weight offset and the slot in L), clamped at 0;
//   2. suffix sums over u: one column per lane, rows walked downwards as in the reference;
//   3. forward substitution: row j is a dot product over lanes + wave butterfly reduction
//      (the only place where the fp32 summation order differs from the reference's serial
//      loop), followed by the reference's /Tdown, clamp and 1e-20 rescaling;
//   4. normalisation (wave reduction) and emission: one frequency per lane, serial over the
//      enthalpy bins from Ibeg[f] exactly as the reference.
// Algorithmic HBM bytes per cell and size: 4*NFREQ in + 4*NFREQ out (tables are L2-resident).
//
// EqTemperature (kernel_A2E.c:110-154) is one lane per cell, operation for operation.
#include "soc_dev.h"
#include "soc_math.h"

__device__ __forceinline__ float soc_wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

#define A2E_IND(a, b) (((a) * (a) - (a)) / 2 + (b))

__global__ __launch_bounds__(64) void soc_a2e_dosolve_kernel(const SocA2EArgs A)
{
    extern __shared__ float lds[];
    const int NE = A.NE, NFREQ = A.NFREQ, lane = threadIdx.x;
    float *L    = lds;                                   // [(NE*NE-NE)/2]
    float *XL   = L + (NE * NE - NE) / 2;                // [NE]
    float *sABS = XL + NE;                               // [NFREQ]
    float *sAF  = sABS + NFREQ;                          // [NFREQ]
    const int cell = blockIdx.x;
    if (cell >= A.batch) return;
    const float *ABS = A.AABS + (size_t)cell * NFREQ;
    for (int i = lane; i < NFREQ; i += 64) { sABS[i] = ABS[i];  sAF[i] = A.AF[i]; }
    __syncthreads();

    // 1. heating: L[u,l] = max(sum_i ABS[i]*Iw*AF[i], 0)   (kernel_A2E.c:45-54).  Four pairs per lane at a time: their
    //    descriptors and weights are asked for together (one wave per SIMD here: nothing else hides the latency of the
    //    table reads); every sum runs over its own window in the reference's order.
    for (int e0 = lane; e0 < A.npair; e0 += 256) {
        int   i0[4], n[4], dst[4];
        const float *w[4];
        float I[4];
        int   nmax = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int e = e0 + 64 * q;
            const bool on = e < A.npair;
            i0[q]  = on ? A.pair_first[e] : 0;
            n[q]   = on ? (A.pair_last[e] - i0[q] + 1) : 0;
            w[q]   = A.Iw + (on ? A.pair_iw[e] : 0);
            dst[q] = on ? A.pair_dst[e] : -1;
            I[q]   = 0.0f;
            nmax   = n[q] > nmax ? n[q] : nmax;
        }
        for (int t = 0; t < nmax; t++) {
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (t < n[q]) I[q] += sABS[i0[q] + t] * w[q][t] * sAF[i0[q] + t];
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (dst[q] >= 0) L[dst[q]] = __builtin_fmaxf(I[q], 0.0f);
    }
    __syncthreads();
    // 2. suffix sums over the upper level (kernel_A2E.c:72-77): a lane owns the columns lane, lane + 64, ... and walks
    //    them together, row by row from the bottom -- per column the reference's order, several chains in flight
    for (int j = NE - 3; j > 0; j--) {
        for (int i = lane; i < j; i += 64) L[A2E_IND(j, i)] += L[A2E_IND(j + 1, i)];
    }
    __syncthreads();
    // 3. forward substitution (kernel_A2E.c:80-88), row per lane: lane r of a block of 64 rows keeps the sum of its row
    //    XL[j] = sum_{i<j} L[j,i] * XL[i], added up in the reference's order (i ascending, mul then add); the rows of a
    //    block are finished one after the other, the finished XL[j] handed to the rows below by v_readlane.  No barrier,
    //    no tree reduction: the same fp32 operations as the serial loop, so the same bits.
    if (lane == 0) XL[0] = 1.0e-20f;
    for (int j0 = 0; j0 < NE; j0 += 64) {
        const int row = j0 + lane;
        const bool live = (row < NE);
        float s = 0.0f;
        if (live) for (int i = 0; i < j0; i++) s += L[A2E_IND(row, i)] * XL[i];       // the blocks above: all XL[i] final
        const int kend = (NE - j0 < 64) ? (NE - j0) : 64;
        const float td = live ? (A.Tdown[row] + 1.0e-30f) : 1.0f;                     // lane k finishes row j0 + k
        for (int k = 0; k < kend; k++) {
            const int jk = j0 + k;
            float x = 1.0e-20f;                                                       // XL[0]
            if (jk > 0) {
                x = s / td;
                x = __builtin_fmaxf(x, 0.0f);
            }
            float xk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), k));
            if (xk > 1.0e20f) {
                // rescaling (kernel_A2E.c:86-88): XL[0..jk] *= 1e-20, and every later row is summed from those values --
                // the sums in hand were taken with the old ones, so they are taken again (rare)
                for (int i = lane; i < jk; i += 64) XL[i] *= 1.0e-20f;
                xk *= 1.0e-20f;
                if (lane == k) XL[jk] = xk;
                s = 0.0f;
                if (live && row > jk) for (int i = 0; i <= jk; i++) s += L[A2E_IND(row, i)] * XL[i];
            } else {
                if (lane == k) XL[jk] = xk;
                if (live && row > jk) s += L[A2E_IND(row, jk)] * xk;
            }
        }
    }
    __syncthreads();
    // normalise (kernel_A2E.c:90-92): the sum in the reference's order, every lane for itself (broadcast reads)
    float nrm = 0.0f;
    for (int i = 0; i < NE; i++) nrm += XL[i];
    nrm = 1.0f / nrm;
    __syncthreads();
    for (int i = lane; i < NE; i += 64) XL[i] = XL[i] * nrm;
    __syncthreads();
    // 4. emission (kernel_A2E.c:95-100): one frequency per lane, serial over the bins
    float *EMIT = A.AEMIT + (size_t)cell * NFREQ;
    for (int f = lane; f < NFREQ; f += 64) {
        float I = 0.0f;
        const float *ea = A.EA + f;                       // EA transposed: [bin][frequency]
        const int ib = A.Ibeg[f];
#pragma unroll 8
        for (int i = ib; i < NE; i++) I += ea[(size_t)i * NFREQ] * XL[i];
        EMIT[f] = I;
    }
}

// EqTemperature (kernel_A2E.c:110-154): trapezoid E_in, log-table lookup of T, Planck emission
__global__ void soc_a2e_eqtemp_kernel(const SocEqTArgs A)
{
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= A.batch) return;
    if (A.icell + id >= A.CELLS) return;
    const float *a = A.ABS + (size_t)id * A.NFREQ;
    float Ein = 0.0f;
    for (int i = 1; i < A.NFREQ; i++)
        Ein += (a[i] * A.FREQ[i] + a[i - 1] * A.FREQ[i - 1]) * ((A.FREQ[i] - A.FREQ[i - 1]) * 3.3130348e-27f);
    int iE = (int)soc_floorf(A.oplgkE * soc_log10f(Ein / A.Emin));
    iE = iE < 0 ? 0 : (iE > A.NIP - 2 ? A.NIP - 2 : iE);
    const float wi = (A.Emin * soc_pownf(A.kE, iE + 1) - Ein) / (A.Emin * soc_pownf(A.kE, iE + 1) - soc_pownf(A.kE, iE));
    const float TP = (float)((double)(wi * A.TTT[iE]) + (1.0 - (double)wi) * (double)A.TTT[iE + 1]);
    A.T[id] = TP;
    for (int f = 0; f < A.NFREQ; f++) {
        const float fr = A.FREQ[f];
        A.EMIT[(size_t)id * A.NFREQ + f] =
            (2.79639459e-20f * A.FACTOR) * A.KABS[f] * (fr * fr / (soc_expf(4.7995074e-11f * fr / TP) - 1.0f));
    }
}

// kernel_eqsolver.c (A2E_MABU.py SolveEquilibriumDust): EqTemperature :5-62 and Emission :66-79 of one equilibrium
// dust component, one lane per cell (A.NIP carries the number of table entries NE)
__global__ void soc_eqsolver_kernel(const SocEqTArgs A)
{
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= A.batch) return;
    if (A.icell + id >= A.CELLS) return;
    const float *a = A.ABS + (size_t)id * A.NFREQ;
    const int   NF = A.NFREQ;
    const float scale = 6.62607e-27f;
    float Ein = 0.0f;
    Ein += a[0] * A.FREQ[0] * scale * (A.FREQ[1] - A.FREQ[0]);
    Ein += a[NF - 1] * A.FREQ[NF - 1] * scale * (A.FREQ[NF - 1] - A.FREQ[NF - 2]);
    for (int i = 1; i < NF - 1; i++) Ein += a[i] * A.FREQ[i] * scale * (A.FREQ[i + 1] - A.FREQ[i - 1]);
    int iE = (int)soc_floorf(A.oplgkE * soc_log10f((0.5f * Ein / 1.0f) / A.Emin));
    iE = iE < 0 ? 0 : (iE > A.NIP - 2 ? A.NIP - 2 : iE);
    const float wi = (A.Emin * soc_pownf(A.kE, iE + 1) - (Ein / 1.0f)) / (A.Emin * soc_pownf(A.kE, iE + 1) - soc_pownf(A.kE, iE));
    float TP = (float)((double)(wi * A.TTT[iE]) + (1.0 - (double)wi) * (double)A.TTT[iE + 1]);
    if (Ein <= 0.0f) TP = 2.7f;
    A.T[id] = TP;
    for (int f = 0; f < NF; f++) {
        const float fr = A.FREQ[f];
        const float res = (2.79639459e-20f * A.FACTOR) * A.KABS[f] * (fr * fr / (soc_expf(4.7995074e-11f * fr / TP) - 1.0f));
        A.EMIT[(size_t)id * NF + f] = (res - res == 0.0f) ? res : 0.0f;          // isfinite
    }
}

hipError_t soc_launch_eqsolver(const SocEqTArgs &A, hipStream_t st)
{
    if (A.batch <= 0) return hipSuccess;
    soc_eqsolver_kernel<<<(A.batch + 255) / 256, 256, 0, st>>>(A);
    return hipGetLastError();
}

hipError_t soc_launch_a2e_dosolve(const SocA2EArgs &A, hipStream_t st)
{
    if (A.batch <= 0) return hipSuccess;
    const size_t lds = (size_t)((A.NE * A.NE - A.NE) / 2 + A.NE + 2 * A.NFREQ) * 4;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)soc_a2e_dosolve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    soc_a2e_dosolve_kernel<<<A.batch, 64, lds, st>>>(A);
    return hipGetLastError();
}

hipError_t soc_launch_a2e_eqtemp(const SocEqTArgs &A, hipStream_t st)
{
    if (A.batch <= 0) return hipSuccess;
    soc_a2e_eqtemp_kernel<<<(A.batch + 255) / 256, 256, 0, st>>>(A);
    return hipGetLastError();
}
