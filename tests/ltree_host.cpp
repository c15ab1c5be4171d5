// ltree_host.cpp -- host harness for soc_amd/csrc/soc_ltree.h + soc_lbricks.h (TEST CODE): follows rays through the
// brick-local hierarchies exactly as the device walk does (GetStep's float arithmetic, then soc_lt_aim / soc_lt_land)
// and reports every step, so that tests/test_ltree.py can compare it with the oracle's IndexG/GetStep/Index
// (oracle/soc_oracle.c, double Index) step by step.  Built by tests/util.py with g++ (optionally with sanitizers).
#include <cmath>
#include <cstdio>
#include <vector>

#include "../soc_amd/csrc/soc_lbricks.h"
#include "../soc_amd/csrc/soc_octbricks.h"

#define PEPS 1.0e-4f

struct Harness {
    int NX, NY, NZ, LEVELS;
    std::vector<int> LCELLS, OFF;
    SocLBricksHost B;
};

extern "C" {

void *lt_build(int NX, int NY, int NZ, int LEVELS, const int *LCELLS, const int *OFF, const float *DENS, int cap)
{
    Harness *H = new Harness();
    H->NX = NX;  H->NY = NY;  H->NZ = NZ;  H->LEVELS = LEVELS;
    H->LCELLS.assign(LCELLS, LCELLS + LEVELS);
    H->OFF.assign(OFF, OFF + LEVELS);
    if (!soc_lbricks_build(NX, NY, NZ, LEVELS, LCELLS, OFF, DENS, cap, H->B)) { delete H;  return nullptr; }
    return H;
}

void lt_free(void *h) { delete (Harness *)h; }

int lt_info(void *h, int *nbricks, int *max_slots, long *nslots)
{
    Harness *H = (Harness *)h;
    *nbricks = (int)H->B.bricks.size();  *max_slots = H->B.max_slots;  *nslots = (long)H->B.btree.size();
    return 0;
}

// consistency of the bricks with the hierarchy: every cell has exactly one slot, links and densities are those of DENS
int lt_check(void *h, const float *DENS, long cells)
{
    Harness *H = (Harness *)h;
    std::vector<unsigned char> seen((size_t)cells, 0);
    for (size_t b = 0; b < H->B.bricks.size(); b++) {
        const SocLBrick &K = H->B.bricks[b];
        for (int s = 0; s < K.nslot; s++) {
            const int g = H->B.bcell[K.base + s];
            if (g < 0 || g >= cells || seen[g]) return -1;
            seen[g] = 1;
            const float d = DENS[g], t = H->B.btree[K.base + s];
            if (d > 0.0f) { if (t != d) return -2; }
            else {
                if (t > 0.0f) return -3;
                const int ls = soc_lt_link(t);
                if (ls < 0 || ls + 8 > K.nslot) return -4;
                int lev = 0;
                while (lev + 1 < H->LEVELS && g >= H->OFF[lev + 1]) lev++;
                const int child = H->OFF[lev + 1] + soc_lb::link_of(d);
                for (int k = 0; k < 8; k++) if (H->B.bcell[K.base + ls + k] != child + k) return -5;
            }
        }
    }
    for (long i = 0; i < cells; i++) if (!seen[i]) return -6;
    return 0;
}

// Follow one ray through soc_lt_move(), the single-path form the device walk uses: steps, arrivals in the next brick and --
// after every move -- a placement of the packet's own cell, which must find the slot and the density the move reported.
// Per step: level, global cell index, step length (root units), as orc_trace reports them.  Returns the number of steps;
// *status: 0 left the model, 1 maxsteps, 2 stopped at a step that needs the generic Index, -1 lost, -3 a placement disagreed.
int lt_trace_move(void *h, const float *pos, const float *dir, int maxsteps, int *levels, int *cells, float *dss, float *endpos, int *status)
{
    Harness *H = (Harness *)h;
    const int NX = H->NX, NY = H->NY, NZ = H->NZ, Lmax = H->LEVELS - 1;
    int k = 1;
    const int nmax = NX > NY ? (NX > NZ ? NX : NZ) : (NY > NZ ? NY : NZ);
    while ((1 << k) <= nmax) k++;
    const int kexp = k - 30;
    float px = pos[0], py = pos[1], pz = pos[2];
    const float ux = dir[0], uy = dir[1], uz = dir[2];
    int level = 0, cx = 0, cy = 0, cz = 0, slot = -1, obase = -1, n = 0, Rx, Ry, Rz;
    float dens = 0.0f;
    *status = 0;
    endpos[0] = px;  endpos[1] = py;  endpos[2] = pz;
    if ((px <= 0.0f) || (py <= 0.0f) || (pz <= 0.0f) || (px >= NX) || (py >= NY) || (pz >= NZ)) return 0;      // IndexG
    int brick = H->B.rbrick[((int)floorf(pz) * NY + (int)floorf(py)) * NX + (int)floorf(px)];
    int what = SOC_LTM_ARRIVE;                              // IndexG = an arrival of a root-level packet
    while (true) {
        const SocLBrick &K = H->B.bricks[brick];
        const float *tree = H->B.btree.data() + K.base;
        if (what == SOC_LTM_STEP) {
            if (n >= maxsteps) { *status = 1;  break; }
            levels[n] = level;
            cells[n]  = H->B.bcell[K.base + slot];
            const float ax = (ux > 0.0f) ? (((1.0f + PEPS) - soc_fmod1f(px)) / ux) : ((-PEPS - soc_fmod1f(px)) / ux);
            const float ay = (uy > 0.0f) ? (((1.0f + PEPS) - soc_fmod1f(py)) / uy) : ((-PEPS - soc_fmod1f(py)) / uy);
            const float az = (uz > 0.0f) ? (((1.0f + PEPS) - soc_fmod1f(pz)) / uz) : ((-PEPS - soc_fmod1f(pz)) / uz);
            const float s = soc_fminf(ax, soc_fminf(ay, az));
            px += s * ux;  py += s * uy;  pz += s * uz;
            dss[n] = soc_scale_down(s, level);
            n++;
        }
        const int L0 = level, c0x = cx, c0y = cy, c0z = cz;
        const int r = soc_lt_move(tree, K, NX, NY, NZ, Lmax, kexp, what, px, py, pz, level, cx, cy, cz, slot, obase, dens, Rx, Ry, Rz);
        if (r == SOC_LT_EXIT) {
            if (L0 > 0) {      // Index() leaves the root-grid position behind (kernel_ASOC_aux.c:238-241); the device walk has no use for it
                const float sc = soc_lt_pow2(-L0);
                px = SOC_FMA(px, sc, (float)(c0x & ~1) * sc);  py = SOC_FMA(py, sc, (float)(c0y & ~1) * sc);  pz = SOC_FMA(pz, sc, (float)(c0z & ~1) * sc);
            }
            break;
        }
        if (r == SOC_LT_SLOW) { *status = 2;  break; }
        if (r == SOC_LT_LOST) { *status = -1;  break; }
        if (r == SOC_LT_LEAVE) {
            brick = H->B.rbrick[(Rz * NY + Ry) * NX + Rx];
            what = SOC_LTM_ARRIVE;
            slot = obase = -1;                                 // (they mean nothing in the next brick)
            continue;
        }
        {   // placement of the cell just found
            int s2 = -1, b2 = -1, l2 = level, ax2 = cx, ay2 = cy, az2 = cz, qx, qy, qz;
            float d2 = 0.0f, p2x = px, p2y = py, p2z = pz;
            const int r2 = soc_lt_move(tree, K, NX, NY, NZ, Lmax, kexp, SOC_LTM_PLACE, p2x, p2y, p2z, l2, ax2, ay2, az2, s2, b2, d2, qx, qy, qz);
            if (r2 != SOC_LT_INSIDE || s2 != slot || d2 != dens || l2 != level || p2x != px || (level > 0 && b2 != obase)) { *status = -3;  break; }
        }
        what = SOC_LTM_STEP;
    }
    endpos[0] = px;  endpos[1] = py;  endpos[2] = pz;
    return n;
}

// soc_octbricks.h (bricks of the sweep that reads the hierarchy from global memory): build, then check what the device
// relies on -- every cell owns exactly one (brick, slot) word, slots of a brick are dense and below 2^SOC_SLOT_BITS,
// a brick exceeds `cap` cells only when it is a single refined cell standing for an oversized subtree's head,
// bcell lists the cells in brick order.  Returns 0 or the number of the violated check; nbricks / largest brick out.
int ob_check(int NX, int NY, int NZ, int LEVELS, const int *LCELLS, const int *OFF, const float *DENS, int cap, int *nbricks, int *largest)
{
    int cells = 0;
    for (int l = 0; l < LEVELS; l++) cells += LCELLS[l];
    SocOctBuilder B(NX, NY, NZ, LEVELS, cells, LCELLS, OFF, DENS, cap);
    B.build();
    const int NB = (int)B.bbase.size() - 1;
    *nbricks = NB;
    *largest = 0;
    if (NB < 1 || B.bbase[0] != 0 || B.bbase[NB] != (int)B.bcell.size()) return 1;
    if ((long)B.bcell.size() != (long)cells) return 2;                       // every cell, refined ones included, holds a slot
    std::vector<char> seen((size_t)cells, 0);
    for (int b = 0; b < NB; b++) {
        const int n = B.bbase[b + 1] - B.bbase[b];
        if (n < 1 || n > cap || n > (int)(SOC_SLOT_MASK + 1u)) return 3;
        if (n > *largest) *largest = n;
        for (int k = 0; k < n; k++) {
            const int a = B.bcell[(size_t)B.bbase[b] + k];
            if (a < 0 || a >= cells || seen[a]) return 4;
            seen[a] = 1;
            if (B.slotmap[a] != (((uint32_t)b << SOC_SLOT_BITS) | (uint32_t)k)) return 5;
        }
    }
    return 0;
}

}  // extern "C"
