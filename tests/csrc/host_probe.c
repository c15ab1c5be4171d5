/* host_probe.c -- host build of the PRODUCT headers (soc_math.h, soc_rng.h) so that their
 * logic can be tested without a GPU.  Compiled by tests/hostprobe.py with gcc. */
#include <stdint.h>
#include "../../soc_amd/csrc/soc_math.h"
#include "../../soc_amd/csrc/soc_rng.h"

uint64_t hp_seed_base(float SEED) { return soc_seed_base(SEED); }
uint64_t hp_seed_mul(float SEED) { return soc_seed_mul(SEED); }
void hp_build_table(uint64_t *tab) { soc_build_seed_table(tab); }
uint64_t hp_mulmod(uint64_t a, uint64_t b) { return soc_mulmod(a, b); }
uint64_t hp_powmod(uint64_t a, uint64_t e) { return soc_powmod(a, e); }

void hp_seed_stream(float SEED, const uint64_t *tab, uint32_t gid, uint32_t *x, uint32_t *c)
{
    soc_rng_t s = soc_seed_stream(soc_seed_mul(SEED), tab, gid);
    *x = s.x;
    *c = s.c;
}

void hp_draws(uint32_t *x, uint32_t *c, int n, uint32_t *u, float *r)
{
    soc_rng_t s = { *x, *c };
    for (int i = 0; i < n; i++) {
        soc_rng_t t = s;
        u[i] = soc_next_uint(&s);
        r[i] = soc_rand(&t);
    }
    *x = s.x;
    *c = s.c;
}

void hp_math(int fn, const float *x, float *y, long n)
{
    for (long i = 0; i < n; i++) {
        switch (fn) {
        case 0: y[i] = soc_expf(x[i]); break;
        case 1: y[i] = soc_logf(x[i]); break;
        case 2: y[i] = soc_sinf(x[i]); break;
        case 3: y[i] = soc_cosf(x[i]); break;
        case 4: y[i] = soc_acosf(x[i]); break;
        case 5: y[i] = soc_sqrtf(x[i]); break;
        case 6: y[i] = soc_fmod1f(x[i]); break;
        case 8: y[i] = soc_expm1f(x[i]); break;
        case 11: y[i] = soc_expf_small(x[i]); break;
        case 9: y[i] = soc_pow15f(x[i]); break;
        case 10: y[i] = (float)soc_logd((double)x[i]); break;
        default: y[i] = 0.0f;
        }
    }
}

void hp_div_by_rcp(const float *n, const float *u, float *q, long cnt)
{
    for (long i = 0; i < cnt; i++) q[i] = soc_div_by_rcp(n[i], u[i], 1.0f / u[i]);
}

void hp_atan2(const float *y, const float *x, float *r, long cnt)
{
    for (long i = 0; i < cnt; i++) r[i] = soc_atan2f(y[i], x[i]);
}
