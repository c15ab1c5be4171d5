// ref_sca.cpp -- driver for the reference's kernel_ASOC_sca.c (scattered-light kernels)
// compiled unmodified for x86-64 by oracle/build.py.  TEST INFRASTRUCTURE ONLY.
#include "ref_builtins.inc"

typedef int int2 __attribute__((ext_vector_type(2)));

extern "C" {
void zero_out(int NDIR, int2 NPIX, float *OUT);
void SimRAM_PB(int SOURCE, int PACKETS, int BATCH, float SEED, float *ABS, float *SCA, float BG, float3 *PSPOS, float *PS,
               int *LCELLS, int *OFF, int *PAR, float *DENS, float *DSC, float *CSC, int NDIR, float3 *ODIRS, int2 NPIX,
               float MAP_DX, float3 CENTRE, float3 *ORA, float3 *ODE, float *OUT, float *ABU, float *OPT,
               float *XPS_NSIDE, float *XPS_SIDE, float *XPS_AREA, int *ROI_DIM, float *ROI_LOAD);
void SimRAM_HP(int PACKETS, int BATCH, float SEED, float *ABS, float *SCA, int *LCELLS, int *OFF, int *PAR, float *DENS,
               float *DSC, float *CSC, int NDIR, float3 *ODIRS, int2 NPIX, float MAP_DX, float3 CENTRE, float3 *ORA,
               float3 *ODE, float *OUT, float *ABU, float *OPT, float *BG, float *HPBGP);
void SimRAM_PS(int PACKETS, int BATCH, float SEED, float *ABS, float *SCA, float BG, float3 *PSPOS, float *PS,
               int *LCELLS, int *OFF, int *PAR, float *DENS, float *DSC, float *CSC, int NDIR, float3 *ODIRS, int2 NPIX,
               float MAP_DX, float3 CENTRE, float3 *ORA, float3 *ODE, float *OUT, float *ABU, float *OPT,
               float *XPS_NSIDE, float *XPS_SIDE, float *XPS_AREA);
void SimRAM_CL(int SOURCE, int PACKETS, int BATCH, float SEED, float *ABS, float *SCA, int *LCELLS, int *OFF, int *PAR,
               float *DENS, float *EMIT, float *DSC, float *CSC, int NDIR, float3 *ODIRS, int2 NPIX, float MAP_DX,
               float3 CENTRE, float3 *ORA, float3 *ODE, float *OUT, float *OPT, float *ABU, float *EMWEI);
void Parents(float *DENS, int *LCELLS, int *OFF, int *PAR);

struct sca_args {
    int   SOURCE, PACKETS, BATCH, GLOBAL, NDIR, NPIX_X, NPIX_Y;
    float SEED, BG, MAP_DX, CX, CY, CZ;
    float *ABS, *SCA, *PSPOS, *PS;
    int   *LCELLS, *OFF, *PAR;
    float *DENS, *EMIT, *DSC, *CSC, *ODIRS, *ORA, *ODE, *OUT, *OPT, *EMWEI;
    int   *XPS_NSIDE, *XPS_SIDE;      // int32 on the host, float* in the kernel (as in ASOCS.py)
    float *XPS_AREA;
    float *HPBG, *HPBGP;              // Healpix sky of the current frequency (sca SimRAM_HP)
    float *ABU;                       // -D WITH_MSF: ABU[CELLS*NDUST]; ABS, SCA, DSC, CSC then hold NDUST entries / tables
};

// kind 0: SimRAM_PB, 1: SimRAM_CL, 2: SimRAM_PS, 3: SimRAM_HP; work items gid0, gid0+stride, ... < gid1
void ref_sca_sim(const sca_args *a, int kind, int gid0, int gid1, int stride)
{
    float dummy[8] = { 0 };
    int   idummy[8] = { 0 };
    int2  NPIX;  NPIX.x = a->NPIX_X;  NPIX.y = a->NPIX_Y;
    float3 C;    C.x = a->CX;  C.y = a->CY;  C.z = a->CZ;
    g_gsize = (size_t)a->GLOBAL;
    if (stride < 1) stride = 1;
    for (int id = gid0; id < gid1; id += stride) {
        g_gid = (size_t)id;
        if (kind == 0)
            SimRAM_PB(a->SOURCE, a->PACKETS, a->BATCH, a->SEED, a->ABS, a->SCA, a->BG, (float3 *)a->PSPOS, a->PS,
                      a->LCELLS, a->OFF, a->PAR, a->DENS, a->DSC, a->CSC, a->NDIR, (float3 *)a->ODIRS, NPIX, a->MAP_DX, C,
                      (float3 *)a->ORA, (float3 *)a->ODE, a->OUT, a->ABU ? a->ABU : dummy, a->OPT ? a->OPT : dummy, (float *)a->XPS_NSIDE,
                      (float *)a->XPS_SIDE, a->XPS_AREA, idummy, dummy);
        else if (kind == 1)
            SimRAM_CL(a->SOURCE, a->PACKETS, a->BATCH, a->SEED, a->ABS, a->SCA, a->LCELLS, a->OFF, a->PAR, a->DENS, a->EMIT,
                      a->DSC, a->CSC, a->NDIR, (float3 *)a->ODIRS, NPIX, a->MAP_DX, C, (float3 *)a->ORA, (float3 *)a->ODE,
                      a->OUT, a->OPT ? a->OPT : dummy, a->ABU ? a->ABU : dummy, a->EMWEI ? a->EMWEI : dummy);
        else if (kind == 3)
            SimRAM_HP(a->PACKETS, a->BATCH, a->SEED, a->ABS, a->SCA, a->LCELLS, a->OFF, a->PAR, a->DENS, a->DSC, a->CSC,
                      a->NDIR, (float3 *)a->ODIRS, NPIX, a->MAP_DX, C, (float3 *)a->ORA, (float3 *)a->ODE, a->OUT, a->ABU ? a->ABU : dummy,
                      a->OPT ? a->OPT : dummy, a->HPBG, a->HPBGP ? a->HPBGP : dummy);
        else
            SimRAM_PS(a->PACKETS, a->BATCH, a->SEED, a->ABS, a->SCA, a->BG, (float3 *)a->PSPOS, a->PS, a->LCELLS, a->OFF,
                      a->PAR, a->DENS, a->DSC, a->CSC, a->NDIR, (float3 *)a->ODIRS, NPIX, a->MAP_DX, C, (float3 *)a->ORA,
                      (float3 *)a->ODE, a->OUT, a->ABU ? a->ABU : dummy, a->OPT ? a->OPT : dummy, (float *)a->XPS_NSIDE, (float *)a->XPS_SIDE,
                      a->XPS_AREA);
    }
}

void ref_sca_parents(float *DENS, int *LCELLS, int *OFF, int *PAR)
{
    g_gid = 0;  g_gsize = 1;
    Parents(DENS, LCELLS, OFF, PAR);
}
}
