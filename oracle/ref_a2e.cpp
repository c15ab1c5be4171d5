// ref_a2e.cpp -- driver for the reference's kernel_A2E.c (DoSolve, EqTemperature) compiled
// unmodified for x86-64 by oracle/build.py.  TEST INFRASTRUCTURE ONLY.
// Work groups are executed one after another: get_group_id = g, get_local_id = l,
// get_global_id = g*LOCAL + l (LOCAL is the -D LOCAL the object was built with).
#include "ref_builtins.inc"

extern "C" {
void DoSolve(int batch, int isize, float *Iw, int *L1, int *L2, float *Tdown, float *EA, int *Ibeg, float *AF,
             float *AABS, float *AEMIT, float *LL);
void EqTemperature(int icell, float kE, float oplgkE, float Emin, float *FREQ, float *KABS, float *TTT,
                   float *ABS, float *T, float *EMIT);

// GLOBAL work items in groups of LOCAL; LL must hold GLOBAL*(NE*NE-NE)/2 floats
void ref_dosolve(int GLOBAL, int LOCAL, int batch, int isize, float *Iw, int *L1, int *L2, float *Tdown, float *EA,
                 int *Ibeg, float *AF, float *AABS, float *AEMIT, float *LL)
{
    g_gsize = (size_t)GLOBAL;
    for (int id = 0; id < GLOBAL; id++) {
        g_gid = (size_t)id;
        g_ref_lid = (size_t)(id % LOCAL);
        g_ref_wg = (size_t)(id / LOCAL);
        DoSolve(batch, isize, Iw, L1, L2, Tdown, EA, Ibeg, AF, AABS, AEMIT, LL);
    }
}

void ref_eqtemp(int GLOBAL, int icell, float kE, float oplgkE, float Emin, float *FREQ, float *KABS, float *TTT,
                float *ABS, float *T, float *EMIT)
{
    g_gsize = (size_t)GLOBAL;
    for (int id = 0; id < GLOBAL; id++) {
        g_gid = (size_t)id;
        EqTemperature(icell, kE, oplgkE, Emin, FREQ, KABS, TTT, ABS, T, EMIT);
    }
}
}
