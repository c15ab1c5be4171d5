// ref_map.cpp -- driver for the reference's kernel_ASOC_map.c (map-making kernels) compiled
// unmodified for x86-64 by oracle/build.py.  TEST INFRASTRUCTURE ONLY.
#include "ref_builtins.inc"
#ifndef ROI_MAP
#define ROI_MAP 0
#endif

typedef int int2 __attribute__((ext_vector_type(2)));

extern "C" {
void Mapping(float MAP_DX, int2 NPIX, float *MAP, float *EMIT, float3 DIR, float3 RA, float3 DE, int *LCELLS, int *OFF,
             int *PAR, float *DENS, float ABS, float SCA, float3 CENTRE, float3 INTOBS, float *OPT, float *SAVETAU,
             int SAVE_COLDEN
#if (ROI_MAP > 0)
             , int *ROI
#endif
             );
void HealpixMapping(float MAP_DX, int2 NPIX, float *MAP, float *EMIT, float3 DIR, float3 RA, float3 DE, int *LCELLS,
                    int *OFF, int *PAR, float *DENS, float ABS, float SCA, float3 CENTRE, float3 INTOBS, float *OPT,
                    float *SAVETAU, int SAVE_COLDEN
#if (ROI_MAP > 0)
                    , int *ROI
#endif
                    );

void PSTau(int no, float3 *PSPOS, float3 DIR, float3 RA, float3 DE, int *LCELLS, int *OFF, int *PAR, float *DENS, float ABS,
           float SCA, float *OPT, float *pscolden, float *pstau);

struct map_args {
    int   NPIX_X, NPIX_Y, SAVE_COLDEN, healpix;
    float MAP_DX, ABS, SCA;
    float DIR[4], RA[4], DE[4], CENTRE[4], INTOBS[4];
    int   *LCELLS, *OFF, *PAR;
    float *DENS, *EMIT, *OPT, *MAP, *SAVETAU;
    int   *ROI;                       // -D ROI_MAP builds: [x0,x1,y0,y1,z0,z1]
};

static float3 f3of(const float *p) { float3 v;  v.x = p[0];  v.y = p[1];  v.z = p[2];  return v; }

// all pixels (work items) of one map
void ref_map(const map_args *a, int npixels)
{
    float dummy[8] = { 0 };
    int2  NPIX;  NPIX.x = a->NPIX_X;  NPIX.y = a->NPIX_Y;
    g_gsize = (size_t)npixels;
    for (int id = 0; id < npixels; id++) {
        g_gid = (size_t)id;
        if (a->healpix)
            HealpixMapping(a->MAP_DX, NPIX, a->MAP, a->EMIT, f3of(a->DIR), f3of(a->RA), f3of(a->DE), a->LCELLS, a->OFF, a->PAR,
                           a->DENS, a->ABS, a->SCA, f3of(a->CENTRE), f3of(a->INTOBS), a->OPT ? a->OPT : dummy, a->SAVETAU,
                           a->SAVE_COLDEN
#if (ROI_MAP > 0)
                           , a->ROI
#endif
                           );
        else
            Mapping(a->MAP_DX, NPIX, a->MAP, a->EMIT, f3of(a->DIR), f3of(a->RA), f3of(a->DE), a->LCELLS, a->OFF, a->PAR, a->DENS,
                    a->ABS, a->SCA, f3of(a->CENTRE), f3of(a->INTOBS), a->OPT ? a->OPT : dummy, a->SAVETAU, a->SAVE_COLDEN
#if (ROI_MAP > 0)
                    , a->ROI
#endif
                    );
    }
}

// PSTau: one work item per point source; PSPOS = cl float3 array (16 bytes per source)
void ref_pstau(const map_args *a, int no, float *PSPOS, float *pscolden, float *pstau)
{
    float dummy[8] = { 0 };
    g_gsize = (size_t)no;
    for (int id = 0; id < no; id++) {
        g_gid = (size_t)id;
        PSTau(no, (float3 *)PSPOS, f3of(a->DIR), f3of(a->RA), f3of(a->DE), a->LCELLS, a->OFF, a->PAR, a->DENS, a->ABS, a->SCA,
              a->OPT ? a->OPT : dummy, pscolden, pstau);
    }
}
}
