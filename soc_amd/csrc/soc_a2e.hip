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

    // 1. heating: L[u,l] = max(sum_i ABS[i]*Iw*AF[i], 0)   (kernel_A2E.c:45-54)
    for (int e = lane; e < A.npair; e += 64) {
        const int i0 = A.pair_first[e], i1 = A.pair_last[e];
        const float *w = A.Iw + A.pair_iw[e];
        float I = 0.0f;
        for (int i = i0; i <= i1; i++) I += sABS[i] * w[i - i0] * sAF[i];
        L[A.pair_dst[e]] = __builtin_fmaxf(I, 0.0f);
    }
    __syncthreads();
    // 2. suffix sums over the upper level, column by column (kernel_A2E.c:72-77)
    for (int i = lane; i < NE - 2; i += 64) {
        for (int j = NE - 3; j > i; j--) L[A2E_IND(j, i)] += L[A2E_IND(j + 1, i)];
    }
    __syncthreads();
    // 3. forward substitution (kernel_A2E.c:80-88)
    if (lane == 0) XL[0] = 1.0e-20f;
    __syncthreads();
    for (int j = 1; j < NE; j++) {
        float part = 0.0f;
        for (int i = lane; i < j; i += 64) part += L[A2E_IND(j, i)] * XL[i];
        float x = soc_wave_sum(part);
        x = x / (A.Tdown[j] + 1.0e-30f);
        x = __builtin_fmaxf(x, 0.0f);
        if (x > 1.0e20f) {                               // uniform in the wave
            for (int i = lane; i < j; i += 64) XL[i] *= 1.0e-20f;
            x *= 1.0e-20f;
        }
        if (lane == 0) XL[j] = x;
        __syncthreads();
    }
    // normalise (kernel_A2E.c:90-92)
    float s = 0.0f;
    for (int i = lane; i < NE; i += 64) s += XL[i];
    s = 1.0f / soc_wave_sum(s);
    for (int i = lane; i < NE; i += 64) XL[i] = XL[i] * s;
    __syncthreads();
    // 4. emission (kernel_A2E.c:95-100): one frequency per lane, serial over the bins
    float *EMIT = A.AEMIT + (size_t)cell * NFREQ;
    for (int f = lane; f < NFREQ; f += 64) {
        float I = 0.0f;
        const float *ea = A.EA + (size_t)f * NE;
        for (int i = A.Ibeg[f]; i < NE; i++) I += ea[i] * XL[i];
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
