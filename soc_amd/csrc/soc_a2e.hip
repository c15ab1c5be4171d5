// soc_a2e.hip -- stochastically heated grains: DoSolve and EqTemperature of kernel_A2E.c for gfx950.
//
// Reference shape (kernel_A2E.c:2-104): one work item per cell, the lower-triangular
// transition matrix L[(NE^2-NE)/2] of every cell spilled to a global scratch buffer
// interleaved by LOCAL (NE=128 -> 266 MB per batch of 8192 cells; the source comments put
// most of the run time into building and re-reading it), a private XL[NE].
//
// Here: a workgroup of sixteen waves solves C cells (C = 4 where four matrices fit the CU's LDS: NE <= 128); L lives in LDS
// (32.5 KB per cell at NE=128) and never touches HBM:
//   1. heating rates, by ALL lanes of the workgroup for ALL its cells: the (l,u) pairs are spread over the 1024 lanes; a lane
//      reads a pair's integration weights once and adds them up for its C cells (the weights are the same for every cell:
//      read per cell -- as the first form of this kernel did, one wave per cell -- the 650 KB table of NE=128 came from L2
//      8192 times per batch).  Each entry is the reference's sequential sum over its frequency window (host-built pair
//      tables give the window, the weight offset and the slot in L), clamped at 0;
//   then wave w < C goes on with cell w alone (the other waves wait at the barriers):
//   2. suffix sums over u: a lane owns columns, the running sum in a register, rows walked downwards as in the reference;
//   3. forward substitution, row per lane: a lane adds up its row in the reference's order (i ascending, multiply then
//      add), finished values go down the rows with v_readlane -- no barrier, no tree reduction: the reference's bits;
//   4. normalisation and emission: one frequency per lane, serial over the enthalpy bins from Ibeg[f] exactly as the reference.
// Algorithmic HBM bytes per cell and size: 4*NFREQ in + 4*NFREQ out (tables are L2-resident).
//
// EqTemperature (kernel_A2E.c:110-154) is one lane per cell, operation for operation.
#include "soc_dev.h"
#include "soc_math.h"

#define A2E_IND(a, b) (((a) * (a) - (a)) / 2 + (b))

// cycles per phase of DoSolve for experiments (-DSOC_A2E_PROF; tools/exp_a2e.py prints them): wave 0 of every workgroup
#if defined(SOC_A2E_PROF)
__device__ unsigned long long g_a2e_prof[8];
#define A2E_PROF_DECL unsigned long long pt_ = __builtin_readcyclecounter()
#define A2E_PROF(i) do { const unsigned long long t_ = __builtin_readcyclecounter();  if (threadIdx.x == 0) atomicAdd(&g_a2e_prof[i], t_ - pt_);  pt_ = t_; } while (0)
extern "C" __attribute__((visibility("default"))) void soc_a2e_prof_read(unsigned long long *out, int reset)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_a2e_prof), sizeof(unsigned long long) * 8);
    if (reset) { unsigned long long z[8] = { 0 };  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_a2e_prof), z, sizeof(z)); }
}
#else
#define A2E_PROF_DECL
#define A2E_PROF(i) do { } while (0)
#endif
// s = sum_{i<n} Lr[i] * X[i], i ascending, multiply then add -- the order of the reference's serial loop (kernel_A2E.c:82) -- with the
// LDS reads of eight terms in flight before the first addition needs one
__device__ __forceinline__ float a2e_row_sum(const float *Lr, const float *X, const int n)
{
    float s = 0.0f;
    int i = 0;
    for (; i + 8 <= n; i += 8) {
        float l[8], x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { l[u] = Lr[i + u];  x[u] = X[i + u]; }
#pragma unroll
        for (int u = 0; u < 8; u++) s += l[u] * x[u];
    }
    for (; i < n; i++) s += Lr[i] * X[i];
    return s;
}

#define A2E_T 1024                   // threads per workgroup: sixteen waves build the matrices (their LDS leaves room for one workgroup
                                     // per CU, so the waves that hide the latency of the weight reads must be its own); four of them solve
#define A2E_Q 2                      // pairs a lane has in flight in step 1

template <int C>
__global__ __launch_bounds__(A2E_T) void soc_a2e_dosolve_kernel(const SocA2EArgs A)
{
    extern __shared__ float lds[];
    const int T = (int)blockDim.x;                       // 1024, or 256 where the matrices are small (several workgroups per CU)
    const int NE = A.NE, NFREQ = A.NFREQ, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int LSZ = (NE * NE - NE) / 2;
    float *Lall = lds;                                   // [C][(NE*NE-NE)/2]
    float *XLall = Lall + C * LSZ;                       // [C][NE]
    float *sABS = XLall + C * NE;                        // [C][NFREQ]
    float *sAF  = sABS + C * NFREQ;                      // [NFREQ]
    const int cell0 = blockIdx.x * C;
    const int nc = (A.batch - cell0 < C) ? (A.batch - cell0) : C;          // cells of this workgroup (>= 1)
    for (int i = tid; i < C * NFREQ; i += T) {
        const int c = i / NFREQ;
        sABS[i] = (c < nc) ? A.AABS[(size_t)(cell0 + c) * NFREQ + (i - c * NFREQ)] : 0.0f;
    }
    for (int i = tid; i < NFREQ; i += T) sAF[i] = A.AF[i];
    __syncthreads();
    A2E_PROF_DECL;

    // 1. heating: L[u,l] = max(sum_i ABS[i]*Iw*AF[i], 0)   (kernel_A2E.c:45-54); every sum runs over its own window in the
    //    reference's order, the C cells side by side
    for (int e0 = tid; e0 < A.npair; e0 += T * A2E_Q) {
        int   i0[A2E_Q], n[A2E_Q], dst[A2E_Q];
        const float *w[A2E_Q];
        float I[A2E_Q][C];
        int   nmax = 0;
#pragma unroll
        for (int q = 0; q < A2E_Q; q++) {
            const int e = e0 + T * q;
            const bool on = e < A.npair;
            i0[q]  = on ? A.pair_first[e] : 0;
            n[q]   = on ? (A.pair_last[e] - i0[q] + 1) : 0;
            w[q]   = A.Iw + (on ? A.pair_iw[e] : 0);
            dst[q] = on ? A.pair_dst[e] : -1;
#pragma unroll
            for (int c = 0; c < C; c++) I[q][c] = 0.0f;
            nmax   = n[q] > nmax ? n[q] : nmax;
        }
        for (int t = 0; t < nmax; t++) {
#pragma unroll
            for (int q = 0; q < A2E_Q; q++)
                if (t < n[q]) {
                    const float wt = w[q][t], af = sAF[i0[q] + t];
#pragma unroll
                    for (int c = 0; c < C; c++) I[q][c] += sABS[c * NFREQ + i0[q] + t] * wt * af;
                }
        }
        (void)nmax;
#pragma unroll
        for (int q = 0; q < A2E_Q; q++)
            if (dst[q] >= 0) {
#pragma unroll
                for (int c = 0; c < C; c++) Lall[c * LSZ + dst[q]] = __builtin_fmaxf(I[q][c], 0.0f);
            }
    }
    __syncthreads();
    A2E_PROF(0);                                         // heating rates
    // wave wv goes on with cell wv (a wave without a cell walks along idle: the barriers below are the workgroup's)
    const bool mine = (wv < nc);
    float *L = Lall + (mine ? wv : 0) * LSZ, *XL = XLall + (mine ? wv : 0) * NE;
    // 2. suffix sums over the upper level (kernel_A2E.c:72-77): L[j,i] += L[j+1,i] for j = NE-3 .. 1.  A lane owns the columns
    //    lane, lane + 64, ...; the running sum of a column stays in a register (the reference reads back what it has just
    //    written: the same additions), so the reads of a column do not wait for its writes
    if (mine) {
        for (int i = lane; i < NE - 3; i += 64) {
            float acc = L[A2E_IND(NE - 2, i)];
            int j = NE - 3;
            for (; j - 3 > i; j -= 4) {                                   // four rows of the column read together
                const float a0 = L[A2E_IND(j, i)], a1 = L[A2E_IND(j - 1, i)], a2 = L[A2E_IND(j - 2, i)], a3 = L[A2E_IND(j - 3, i)];
                acc = a0 + acc;  L[A2E_IND(j, i)] = acc;
                acc = a1 + acc;  L[A2E_IND(j - 1, i)] = acc;
                acc = a2 + acc;  L[A2E_IND(j - 2, i)] = acc;
                acc = a3 + acc;  L[A2E_IND(j - 3, i)] = acc;
            }
            for (; j > i; j--) {
                acc = L[A2E_IND(j, i)] + acc;
                L[A2E_IND(j, i)] = acc;
            }
        }
    }
    __syncthreads();
    A2E_PROF(1);                                         // suffix sums
    // 3. forward substitution (kernel_A2E.c:80-88), row per lane: lane r of a block of 64 rows keeps the sum of its row
    //    XL[j] = sum_{i<j} L[j,i] * XL[i], added up in the reference's order (i ascending, mul then add); the rows of a
    //    block are finished one after the other, the finished XL[j] handed to the rows below by v_readlane.  No barrier,
    //    no tree reduction: the same fp32 operations as the serial loop, so the same bits.
    //    Inside a block nothing goes through LDS: a finished value stays in its lane's register (written to XL once, after the
    //    block), and the matrix element of the next row to finish is read before the division of this one is waited for.
    if (mine) {
        for (int j0 = 0; j0 < NE; j0 += 64) {
            const int row = j0 + lane;
            const bool live = (row < NE);
            float s = 0.0f;
            if (live) s = a2e_row_sum(L + A2E_IND(row, 0), XL, j0);                       // the blocks above: all XL[i] final
            const int kend = __builtin_amdgcn_readfirstlane((NE - j0 < 64) ? (NE - j0) : 64);
            const float td = live ? (A.Tdown[row] + 1.0e-30f) : 1.0f;                     // lane k finishes row j0 + k
            const float *Lrow = L + (live ? A2E_IND(row, j0) : 0);                        // L[row, j0 + k], k < lane
            float xmine = 0.0f;                                                           // XL[row], once lane `lane` has finished
            float lnext = (live && lane > 0) ? Lrow[0] : 0.0f;
            for (int k = 0; k < kend; k++) {
                const int jk = j0 + k;
                const float lcur = lnext;
                lnext = (live && lane > k + 1) ? Lrow[k + 1] : 0.0f;                      // (asked for now, used in the next turn)
#if defined(A2E_X_MUL)
                float x = s * td;                                                         // timing experiment only (wrong numbers)
#else
                float x = s / td;
#endif
                x = __builtin_fmaxf(x, 0.0f);
                if (jk == 0) x = 1.0e-20f;                                                // XL[0]
                float xk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), k));
                if (xk > 1.0e20f) {
                    // rescaling (kernel_A2E.c:86-88): XL[0..jk] *= 1e-20, and every later row is summed from those values --
                    // the sums in hand were taken with the old ones, so they are taken again (rare).  The values of this block
                    // that are finished go to LDS first, scaled, so that the sums can be taken from there
                    for (int i = lane; i < j0; i += 64) XL[i] *= 1.0e-20f;
                    xk *= 1.0e-20f;
                    if (lane < k) { xmine *= 1.0e-20f;  XL[row] = xmine; }
                    if (lane == k) { xmine = xk;  XL[jk] = xk; }
                    s = 0.0f;
                    if (live && row > jk) s = a2e_row_sum(L + A2E_IND(row, 0), XL, jk + 1);
                    lnext = (live && lane > k + 1) ? Lrow[k + 1] : 0.0f;
                } else {
                    if (lane == k) xmine = xk;
                    if (live && row > jk) s += lcur * xk;
                }
            }
            if (live) XL[row] = xmine;
        }
    }
    __syncthreads();
    A2E_PROF(2);                                         // forward substitution
    // normalise (kernel_A2E.c:90-92): the sum in the reference's order, every lane for itself (broadcast reads)
    float nrm = 0.0f;
    for (int i = 0; i < NE; i++) nrm += XL[i];
    nrm = 1.0f / nrm;
    __syncthreads();
    if (mine) for (int i = lane; i < NE; i += 64) XL[i] = XL[i] * nrm;
    __syncthreads();
    A2E_PROF(3);                                         // normalisation
    // 4. emission (kernel_A2E.c:95-100): one frequency per lane, serial over the bins
    if (mine) {
        float *EMIT = A.AEMIT + (size_t)(cell0 + wv) * NFREQ;
        for (int f = lane; f < NFREQ; f += 64) {
            float I = 0.0f;
            const float *ea = A.EA + f;                       // EA transposed: [bin][frequency]
            const int ib = A.Ibeg[f];
#pragma unroll 8
            for (int i = ib; i < NE; i++) I += ea[(size_t)i * NFREQ] * XL[i];
            EMIT[f] = A.accumulate ? (EMIT[f] + I) : I;       // (the host's EMITTED += emit of A2E.py:596-600, size after size: the same fp32 additions)
        }
    }
    A2E_PROF(4);                                         // emission
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

template <int C>
static hipError_t a2e_launch(const SocA2EArgs &A, size_t lds, hipStream_t st)
{
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)soc_a2e_dosolve_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    // sixteen waves per workgroup where one workgroup fills the CU's LDS; four where four or more fit (NE <= 64)
    const int T = (lds * 4 <= 160 * 1024) ? 256 : A2E_T;
    soc_a2e_dosolve_kernel<C><<<(A.batch + C - 1) / C, T, lds, st>>>(A);
    return hipGetLastError();
}

hipError_t soc_launch_a2e_dosolve(const SocA2EArgs &A, hipStream_t st)
{
    if (A.batch <= 0) return hipSuccess;
    // cells per workgroup: four (one per wave) where their matrices fit the 160 KB of LDS, else two, else one
    const size_t per_cell = (size_t)((A.NE * A.NE - A.NE) / 2 + A.NE + A.NFREQ) * 4, shared = (size_t)A.NFREQ * 4;
    if (per_cell + shared > 160 * 1024) return hipErrorInvalidValue;
    if (4 * per_cell + shared <= 160 * 1024) return a2e_launch<4>(A, 4 * per_cell + shared, st);
    if (2 * per_cell + shared <= 160 * 1024) return a2e_launch<2>(A, 2 * per_cell + shared, st);
    return a2e_launch<1>(A, per_cell + shared, st);
}

hipError_t soc_launch_a2e_eqtemp(const SocEqTArgs &A, hipStream_t st)
{
    if (A.batch <= 0) return hipSuccess;
    soc_a2e_eqtemp_kernel<<<(A.batch + 255) / 256, 256, 0, st>>>(A);
    return hipGetLastError();
}
