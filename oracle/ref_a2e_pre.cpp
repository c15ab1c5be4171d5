// ref_a2e_pre.cpp -- driver for the two kernels of the reference's kernel_A2E_pre.c that A2E_pre.py uses
// (PrepareIntegrationWeightsTrapezoid :580-736, PrepareTdown :123-212), compiled unmodified for x86-64 by
// oracle/build.py.  TEST INFRASTRUCTURE ONLY.  One work item per lower bin l / upper bin u, one after another.
#include "ref_builtins.inc"

extern "C" {
void PrepareIntegrationWeightsTrapezoid(int NFREQ, int NE, float *Ef, float *E, int *L1, int *L2, float *IW, float *wrk, int *noIw);
void PrepareTdown(int NFREQ, float *FREQ, float *Ef, float *SKABS, int NE, float *E, float *T, float *Tdown);

// GLOBAL as in A2E_pre.py:47: (NE/64+1)*64 work items
void ref_pre_weights(int GLOBAL, int NFREQ, int NE, float *Ef, float *E, int *L1, int *L2, float *IW, float *wrk, int *noIw)
{
    g_gsize = (size_t)GLOBAL;
    for (int id = 0; id < GLOBAL; id++) {
        g_gid = (size_t)id;
        PrepareIntegrationWeightsTrapezoid(NFREQ, NE, Ef, E, L1, L2, IW, wrk, noIw);
    }
}

void ref_pre_tdown(int GLOBAL, int NFREQ, float *FREQ, float *Ef, float *SKABS, int NE, float *E, float *T, float *Tdown)
{
    g_gsize = (size_t)GLOBAL;
    for (int id = 0; id < GLOBAL; id++) {
        g_gid = (size_t)id;
        PrepareTdown(NFREQ, FREQ, Ef, SKABS, NE, E, T, Tdown);
    }
}
}
