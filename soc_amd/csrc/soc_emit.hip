// soc_emit.hip -- equilibrium dust temperature and thermal emission on the device:
// EqTemperature, Emission and Emission2 of kernel_ASOC_aux.c (:745-790, :795-808, :862-888), the
// `CLT` / `CLE` paths of ASOC.py (:2024-2040, :2154-2197).  Pure streaming over the cells; the
// point of having them here is that the absorbed energies never leave the GPU between the
// simulation and the next iteration's emission.
//
// The reference bakes FACTOR (%.4e) and LENGTH = GL*PARSEC (%.5e) into the kernel as float
// literals (ASOC.py:344-362); here they are float arguments that the host rounds the same way.
#include "soc_dev.h"
#include "soc_math.h"

// TNEW[cell] from the absorbed energy per cell (array "EMIT" in the reference), all levels in one launch
__global__ void soc_eqtemp_kernel(const SocGrid G, const float adhoc, const float kE, const float Emin, const int NE,
                                  const float FACTOR, const float LENGTH, const float cr_rate, const float *TTT, const float *EABS, float *TNEW)
{
    __shared__ int sOFF[SOC_MAXL + 1];
    if (threadIdx.x <= SOC_MAXL) sOFF[threadIdx.x] = (threadIdx.x < G.LEVELS) ? G.OFF[threadIdx.x] : G.CELLS;
    __syncthreads();
    const float scale  = (6.62607e-27f * FACTOR) / LENGTH;
    const float oplgkE = 1.0f / soc_log10f(kE);
    const float beta   = 1.0f;
    const long  stride = (long)gridDim.x * blockDim.x;
    for (long ind = (long)blockIdx.x * blockDim.x + threadIdx.x; ind < G.CELLS; ind += stride) {
        int level = 0;
        while (level + 1 < G.LEVELS && ind >= sOFF[level + 1]) level++;
        const float d   = G.DENS[ind];
        float Ein = (scale / adhoc) * EABS[ind] * soc_pownf(8.0f, level) / d;
        if (cr_rate > 0.0f) Ein += 1.0e-27f * FACTOR * cr_rate;           // -D CR_HEATING=1 -D CR_HEATING_RATE (kernel_ASOC_aux.c:769-773)
        int iE = (int)soc_floorf(oplgkE * soc_log10f((Ein / beta) / Emin));
        iE = iE < 0 ? 0 : (iE > NE - 2 ? NE - 2 : iE);
        const float wi = (Emin * soc_pownf(kE, iE + 1) - (Ein / beta)) / (Emin * soc_pownf(kE, iE) * (kE - 1.0f));
        TNEW[ind] = (d > 1.0e-7f) ? soc_clampf(wi * TTT[iE] + (1.0f - wi) * TTT[iE + 1], 3.0f, 1600.0f) : 10.0f;
    }
}

// EMITTED[cell - c0][ifreq] for cells [c0, c1) and nfreq frequencies (Emission2; Emission is nfreq == 1)
__global__ void soc_emission_kernel(const int c0, const int c1, const int nfreq, const float FACTOR, const float LENGTH,
                                    const float *FREQ, const float *FABS, const float *T, float *EMIT)
{
    const long n = (long)(c1 - c0) * nfreq;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
        const int   icell = c0 + (int)(k / nfreq), ifreq = (int)(k % nfreq);
        const float t = T[icell], freq = FREQ[ifreq];
        EMIT[k] = (2.79639459e-20f * FACTOR) * FABS[ifreq] * (freq * freq / (soc_expf(4.7995074e-11f * freq / t) - 1.0f)) / LENGTH;
    }
}

hipError_t soc_launch_eqtemp(const SocGrid &G, float adhoc, float kE, float Emin, int NE, float FACTOR, float LENGTH, float cr_rate,
                             const float *TTT, const float *EABS, float *TNEW, hipStream_t st)
{
    if (G.CELLS <= 0) return hipSuccess;
    const int blocks = (int)(((long)G.CELLS + 255) / 256 < 65536 ? ((long)G.CELLS + 255) / 256 : 65536);
    soc_eqtemp_kernel<<<blocks, 256, 0, st>>>(G, adhoc, kE, Emin, NE, FACTOR, LENGTH, cr_rate, TTT, EABS, TNEW);
    return hipGetLastError();
}

hipError_t soc_launch_emission(int c0, int c1, int nfreq, float FACTOR, float LENGTH, const float *FREQ, const float *FABS,
                               const float *T, float *EMIT, hipStream_t st)
{
    const long n = (long)(c1 - c0) * nfreq;
    if (n <= 0) return hipSuccess;
    const int blocks = (int)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
    soc_emission_kernel<<<blocks, 256, 0, st>>>(c0, c1, nfreq, FACTOR, LENGTH, FREQ, FABS, T, EMIT);
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------
// OPT[CELLS][2] = sum over species of ABU * (AFABS, AFSCA), on the device: the reference builds it on the host
// for every frequency and uploads 8*CELLS bytes (ASOC.py:1146-1160; "0.43 s / 2.5 s", :1177).  Same fp32
// operations in the same order as the numpy expressions, so the values are the host's bit for bit.
// ------------------------------------------------------------------------------------
__global__ void soc_opt_kernel(int cells, int ndust, int single, const float *ABU, const float *AF, float2 *OPT)
{
    // AF = [AFABS[0..ndust), AFSCA[0..ndust)] of the current frequency
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < cells; i += (long)gridDim.x * blockDim.x) {
        float a = 0.0f, s = 0.0f;
        if (single) {                                    // two species with abundances ABU and 1-ABU (ASOC.py:1148-1153)
            const float x = ABU[i], y = 1.0f - x;
            a = a + (x * AF[0] + y * AF[1]);
            s = s + (x * AF[2] + y * AF[3]);
        } else {
            for (int d = 0; d < ndust; d++) {            // ASOC.py:1155-1157
                const float x = ABU[(size_t)i * ndust + d];
                a = a + x * AF[d];
                s = s + x * AF[ndust + d];
            }
        }
        OPT[i] = make_float2(a, s);
    }
}

// -D OPT_IS_HALF: the reference keeps OPT as fp16 (numpy float32 -> float16, round to nearest even, ASOC.py:1158-1159)
// and the kernels widen it again with vload_half (kernel_ASOC_aux.c:12-14).  Widening is exact, so a float array holding
// the fp16-rounded values gives the kernels the same numbers.
__global__ void soc_opt_half_kernel(int cells, float2 *OPT)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < cells; i += (long)gridDim.x * blockDim.x) {
        const float2 o = OPT[i];
        OPT[i] = make_float2((float)(_Float16)o.x, (float)(_Float16)o.y);
    }
}

hipError_t soc_launch_opt_half(int cells, float2 *OPT, hipStream_t st)
{
    if (cells <= 0) return hipSuccess;
    const int blocks = (cells + 255) / 256 < 16384 ? (cells + 255) / 256 : 16384;
    soc_opt_half_kernel<<<blocks, 256, 0, st>>>(cells, OPT);
    return hipGetLastError();
}

hipError_t soc_launch_opt(int cells, int ndust, int single, const float *ABU, const float *AF, float2 *OPT, hipStream_t st)
{
    if (cells <= 0) return hipSuccess;
    const int blocks = (cells + 255) / 256 < 16384 ? (cells + 255) / 256 : 16384;
    soc_opt_kernel<<<blocks, 256, 0, st>>>(cells, ndust, single, ABU, AF, OPT);
    return hipGetLastError();
}
