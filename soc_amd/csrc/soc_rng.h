// soc_rng.h -- MWC64X generator and O(1) stream seeding for gfx950.
//
// Same generator and the same streams as the reference (mwc64x_rng.cl:12-48,
// skip_mwc.cl:64-76, call site kernel_ASOC.c:74-77):
//     state(gid) = split( BASEID * A^(base + gid*2^38)  mod M ),   M = A*2^32 - 1
// The reference evaluates A^dist with a shift-and-add modular exponentiation in every work
// item (~1e5 integer instructions).  Here the host factors the exponent,
//     A^(base + gid*2^38) = A^base * G^gid,        G = A^(2^38) mod M,
// precomputes A^base*BASEID once per launch and four 256-entry tables
//     T[k][b] = G^(b * 256^k) mod M        (independent of the seed, built once),
// and a lane obtains its state with 4 table reads and 4 modular multiplications.
// Modular multiplication uses 2^64 = C64 (mod M) folding instead of shift-and-add.
#ifndef SOC_RNG_H
#define SOC_RNG_H

#include <stdint.h>

#if defined(__HIPCC__)
#  define SOC_RNG_HD __host__ __device__ static inline
#else
#  define SOC_RNG_HD static inline
#endif

#define SOC_MWC_A      4294883355ULL
#define SOC_MWC_M      18446383549859758079ULL
#define SOC_MWC_BASEID 4077358422479273989ULL
#define SOC_MWC_C64    360523849793537ULL      /* 2^64 mod M = (2^32 - A) * 2^32 + 1 */
#define SOC_STREAM_GAP 274877906944ULL         /* 2^38 draws per work item (kernel_ASOC.c:74) */

typedef struct { uint32_t x, c; } soc_rng_t;

SOC_RNG_HD uint64_t soc_umul64hi(uint64_t a, uint64_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

// (a * b) mod M for a, b < M
SOC_RNG_HD uint64_t soc_mulmod(uint64_t a, uint64_t b)
{
    uint64_t lo = a * b;
    uint64_t hi = soc_umul64hi(a, b);
    while (hi != 0) {                         // hi*2^64 + lo == hi*C64 + lo (mod M); <= 6 rounds
        uint64_t tlo = hi * SOC_MWC_C64;
        uint64_t thi = soc_umul64hi(hi, SOC_MWC_C64);
        uint64_t s = lo + tlo;
        hi = thi + (uint64_t)(s < lo);
        lo = s;
    }
    if (lo >= SOC_MWC_M) lo -= SOC_MWC_M;     // M > 2^63: one subtraction suffices
    return lo;
}

SOC_RNG_HD uint64_t soc_powmod(uint64_t a, uint64_t e)
{
    uint64_t sqr = a, acc = 1;
    while (e != 0) {
        if (e & 1) acc = soc_mulmod(acc, sqr);
        sqr = soc_mulmod(sqr, sqr);
        e >>= 1;
    }
    return acc;
}

// state of stream `gid`; seed_mul = BASEID * A^base mod M (host, per launch); T = 4 x 256 table
SOC_RNG_HD soc_rng_t soc_seed_stream(uint64_t seed_mul, const uint64_t *T, uint32_t gid)
{
    uint64_t v = seed_mul;
    uint32_t b0 = gid & 255u, b1 = (gid >> 8) & 255u, b2 = (gid >> 16) & 255u, b3 = gid >> 24;
    if (b0) v = soc_mulmod(v, T[b0]);
    if (b1) v = soc_mulmod(v, T[256 + b1]);
    if (b2) v = soc_mulmod(v, T[512 + b2]);
    if (b3) v = soc_mulmod(v, T[768 + b3]);
    soc_rng_t s;
    s.x = (uint32_t)(v / SOC_MWC_A);
    s.c = (uint32_t)(v % SOC_MWC_A);
    return s;
}

SOC_RNG_HD uint32_t soc_next_uint(soc_rng_t *s)
{
    uint32_t res = s->x ^ s->c;
    uint64_t t = (uint64_t)SOC_MWC_A * s->x + s->c;   // x' = low 32, c' = high 32 (== mad_hi + carry)
    s->x = (uint32_t)t;
    s->c = (uint32_t)(t >> 32);
    return res;
}

// Rand(x) = NextUint / 4294967295.0f (kernel_ASOC_aux.c:127): the divisor rounds to 2^32 in
// fp32, so this is an exact scaling of the rounded uint->float conversion; values lie in [0,1].
SOC_RNG_HD float soc_rand(soc_rng_t *s)
{
    return (float)soc_next_uint(s) * 2.3283064365386963e-10f;
}

// ---- host-side per-launch constants -------------------------------------------------------

// base offset of all streams: (ulong)(fmod(SEED*7.0f*PI,1.0f)*4294967296L), kernel_ASOC.c:77.
// volatile keeps every intermediate in fp32 whatever the host compiler's excess-precision rules.
static inline uint64_t soc_seed_base(float SEED)
{
    volatile float a = SEED * 7.0f;
    volatile float b = a * 3.1415926535897f;
    volatile float f = b - (float)(long long)b;      /* fmod(b, 1.0f): exact for |b| < 2^63 */
    volatile float g = f * 4294967296.0f;
    return (uint64_t)g;
}

// BASEID * A^base mod M
static inline uint64_t soc_seed_mul(float SEED)
{
    return soc_mulmod(SOC_MWC_BASEID, soc_powmod(SOC_MWC_A, soc_seed_base(SEED)));
}

// T[k][b] = G^(b * 256^k) mod M, G = A^(2^38) mod M; tab holds 4*256 entries
static inline void soc_build_seed_table(uint64_t *tab)
{
    uint64_t g = soc_powmod(SOC_MWC_A, SOC_STREAM_GAP);
    for (int k = 0; k < 4; k++) {
        uint64_t acc = 1;
        for (int b = 0; b < 256; b++) {
            tab[256 * k + b] = acc;
            acc = soc_mulmod(acc, g);
        }
        g = acc;                                     /* g^256 */
    }
}

#endif  // SOC_RNG_H
