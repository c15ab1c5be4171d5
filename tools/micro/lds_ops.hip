// LDS micro-benchmark for gfx950: cost of a wave-wide ds_read_b32 / ds_add_f32 / ds_add_u32 with scattered (random) addresses, as the
// brick walk issues them.  Build and run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/micro/lds_ops.hip -o /tmp/lds_ops && /tmp/lds_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CELLS 8704

template <int OP>
__global__ __launch_bounds__(512) void k(float *out, int iters, int spread)
{
    __shared__ float sT[CELLS];
    double *sDd = (double *)sT;
    for (int i = threadIdx.x; i < CELLS; i += blockDim.x) sT[i] = 0.0f;
    __syncthreads();
    uint32_t x = 1234567u + 7919u * (blockIdx.x * blockDim.x + threadIdx.x);
    float acc = 0.0f;
    for (int it = 0; it < iters; it++) {
        x = x * 1664525u + 1013904223u;
        const int a = (int)((x >> 8) % (uint32_t)spread);
        if (OP == 0) acc += sT[a];
        if (OP == 1) atomicAdd(&sT[a], 1.0f);
        if (OP == 2) atomicAdd((unsigned int *)&sT[a], 1u);
        if (OP == 3) { acc += sT[a];  atomicAdd(&sT[(a * 7 + 13) % spread], 1.0f); }
        if (OP == 4) atomicAdd(&sDd[a >> 1], 1.0);
        if (OP == 5) __hip_atomic_fetch_add(&sT[a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (OP == 6) atomicAdd((unsigned long long *)&sDd[a >> 1], 1ull);
        if (OP == 7) { float old = sT[a];  sT[a] = old + 1.0f; }                 // (not atomic: what a read-modify-write costs)
    }
    __syncthreads();
    if (acc == 12345.678f || sT[threadIdx.x % CELLS] == -1.0f) out[0] = acc;
}

template <int OP>
static void run(const char *name, int spread)
{
    float *d;  (void)hipMalloc(&d, 4);
    const int iters = 20000, blocks = 512;      // two 512-thread workgroups per CU on 256 CUs
    hipEvent_t e0, e1;  (void)hipEventCreate(&e0);  (void)hipEventCreate(&e1);
    k<OP><<<blocks, 512>>>(d, 100, spread);
    (void)hipEventRecord(e0);
    k<OP><<<blocks, 512>>>(d, iters, spread);
    (void)hipEventRecord(e1);  (void)hipEventSynchronize(e1);
    float ms = 0;  (void)hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per CU: blocks/256 CUs * 8 waves * iters; cycles at 2.4 GHz
    const double winstr_per_cu = (double)blocks / 256.0 * 8.0 * iters;
    printf("%-34s spread %5d: %8.3f ms  -> %.1f cycles of a CU per wave-wide instruction\n", name, spread, ms, ms * 1e-3 * 2.4e9 / winstr_per_cu);
    (void)hipFree(d);
}

int main()
{
    for (int spread : { CELLS, 64, 1 }) {
        run<0>("ds_read_b32", spread);
        run<1>("ds_add_f32 (no return)", spread);
        run<2>("ds_add_u32 (no return)", spread);
        run<3>("read + float add", spread);
        run<4>("ds_add_f64 (no return)", spread);
        run<5>("float add, workgroup scope", spread);
        run<6>("ds_add_u64 (no return)", spread);
        run<7>("plain read-modify-write (racy)", spread);
    }
    return 0;
}
