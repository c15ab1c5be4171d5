// soc_ltree.h -- neighbour search on a brick-local copy of the hierarchy: Index() (kernel_ASOC_aux.c:198-278) for
// runs that evaluate it in double (NX > DIMLIM), without parent links and without fp64.
//
// The reference keeps a packet's place as (level, ind, pos): pos is local to the octet of the cell, in [0,2]^3 plus
// a small overstep.  After a step Index() climbs -- POS = POS/2 + octant, through PAR -- until POS lies inside an
// octet (or the root grid), takes the cell there and descends with POS = 2*fmod(POS,1) to a leaf; the result is
// rounded to float once, at the end.  With double3 POS every one of those operations is exact as long as no
// coordinate is tiny (below 2^(k+L-30) for a root grid of < 2^k cells and a cell of level L), so what Index()
// returns is
//     the leaf that contains the point, and  pos' = RN_float(pos * 2^(l-L) + B)
// with l the leaf's level and B the (dyadic, exactly representable) distance between the two octet origins in
// units of the new level.  That is what soc_lt_aim() / soc_lt_land() compute, from integers:
//   * the packet carries the integer coordinates (cx, cy, cz) of its cell on the cell's own level instead of ind;
//   * the cell that holds the point has coordinates t = (c & ~1) + floor(pos) on level L, its ancestors t >> j;
//   * the brick's cells sit in LDS (`tree`: a leaf holds its density, a refined cell the link to the slots of its
//     eight children, as in the cloud file but with brick-local slots), root cells of the brick's box first, so the
//     leaf is found by a descent through at most LEVELS reads of LDS;
//   * one fused multiply-add gives pos'.
// Cases in which the reference's own arithmetic is not the exact one -- a coordinate exactly on a cell face (the
// inclusive tests of :264 pick a cell by rule, not by geometry) or tiny -- are recognised (soc_lt_degenerate) and
// left to soc_index<double> itself (the "slow step" queue of the brick sweep).
//
// Shared by the device walk (soc_brick.hip) and the host (tests/ltree_host.cpp follows rays through bricks built
// by soc_lbricks.h and is compared with the oracle's Index step by step).
#ifndef SOC_LTREE_H
#define SOC_LTREE_H

#include "soc_math.h"

struct SocLBrick {
    int x0, y0, z0;            // first root cell of the box
    int bx, by, bz;            // root cells per edge
    int base, nslot;           // first entry in btree / bcell, cells (leaves and refined cells) of the brick
};

enum { SOC_LT_INSIDE = 0, SOC_LT_LEAVE = 1, SOC_LT_EXIT = 2, SOC_LT_SLOW = 3 };

// a * b + c of small non-negative numbers (slots of a brick's box: all below 2^24).  On the device a 24-bit multiply: for the
// plain int expression the compiler picks v_mad_u64_u32, whose 64-bit addend is a register PAIR -- the unused half may be a
// register with a load in flight, and the instruction then waits for ALL loads in flight (seen in the walk: the prefetched
// packet records).
#if defined(__HIP_DEVICE_COMPILE__)
// (b is workgroup-uniform wherever this is used: a scalar register.  Written as the instruction itself: where the compiler
// can bound the operands it turns __mul24 back into a plain multiply and selects the 64-bit form again.)
__device__ __forceinline__ int soc_mad24(int a, int b, int c) { int r;  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));  return r; }
#  define SOC_MAD24(a, b, c) soc_mad24((a), (b), (c))
#else
#  define SOC_MAD24(a, b, c) ((a) * (b) + (c))
#endif

SOC_HD int soc_lt_link(float d) { return (int)(soc_f2u(d) ^ 0x80000000u); }

// exact 2^k for -126 <= k <= 127
SOC_HD float soc_lt_pow2(int k) { return soc_u2f((uint32_t)(127 + k) << 23); }

// Is the step one that Index() (double) would not resolve by exact geometry?  level > 0, pos after the step.
// thr = 2^(k + level - 30): below it pos/2^level + root coordinate has more than 53 significant bits.
SOC_HD bool soc_lt_degenerate(float px, float py, float pz, float flx, float fly, float flz, float thr)
{
    const float m = soc_fminf(soc_fabsf(px), soc_fminf(soc_fabsf(py), soc_fabsf(pz)));
    return !(m >= thr) || (px == flx) || (py == fly) || (pz == flz);
}

// ---------------------------------------------------------------------------------------------------------------
// ONE straight path for every kind of move, so that a wavefront whose lanes make
// different moves (root cell -> root cell, sibling, up, down, into the next brick) executes one instruction stream
// instead of one per kind.
//
// The point reached by the step is written as integer cell coordinates on the FINEST level,
//     F = (o << D) + floor(pos * 2^D),     o = origin of the packet's octet on its level L (0 on the root grid),
//                                           D = Lmax - L,
// which is exact: pos * 2^D is a power-of-two scaling, floor() and the conversion are exact below 2^24, and the
// binary digits of pos - floor(pos) are the octants Index() picks with 2*fmod(POS,1) on the way down (:268-273) --
// also for a negative coordinate, whose fractional part a float subtraction would round.  The ancestors of the
// point's cell are F >> (Lmax - l), so the descent from the brick's root cell (or, for a sibling, from the packet's
// own octet) is one loop with one LDS read per level, and the new local position is one fma.
//
// Two halves, so that the walk can put other work between the first LDS read and its use:
//   soc_lt_aim()   where the point is: F, its root cell R, the outcome if it is already known (LEAVE, EXIT, SLOW) and
//                  the slot `s` (on level `l`) at which the descent starts;
//   soc_lt_land()  given tree[s]: the descent to the leaf and the packet's new place.
// The packet's place is (level, c, slot, obase): obase = slot of the first cell of its octet (level > 0).
//   what == SOC_LTM_STEP   : pos has been advanced by GetStep's arithmetic;
//   what == SOC_LTM_ARRIVE : the same for a packet that comes from another brick (slot and obase mean nothing yet);
//   what == SOC_LTM_PLACE  : find slot, obase and density of the packet's own cell (level, c); pos is not touched.
// Outcomes: SOC_LT_INSIDE (the packet's place is the new leaf), SOC_LT_LEAVE (the point lies in root cell R of another brick: the
// packet keeps its old cell and the advanced pos, the brick of that root cell completes the step), SOC_LT_EXIT (outside the model),
// SOC_LT_SLOW (not a step exact geometry decides) -- the last three change nothing -- or SOC_LT_LOST when a placement does not find the cell
// (a broken record; the walk retires the packet).
// kexp = k - 30 with 2^k > max(NX, NY, NZ): the bounds of soc_lt_degenerate (2^(kexp+L)) and of the sibling case (2^(kexp+1)).
// ---------------------------------------------------------------------------------------------------------------
enum { SOC_LTM_STEP = 0, SOC_LTM_ARRIVE = 1, SOC_LTM_PLACE = 2 };
enum { SOC_LT_LOST = 4 };

struct SocLtAim {
    int Fx, Fy, Fz;            // the point on the finest level
    int Rx, Ry, Rz;            // its root cell
    int ox, oy, oz;            // origin of the packet's octet on its level (0 on the root grid)
    int s, l;                  // the descent starts at slot s, a cell of level l
    int r;                     // SOC_LT_INSIDE: go on with soc_lt_land(tree[s]); else the outcome
};

SOC_HD void soc_lt_aim(const SocLBrick &K, const int NX, const int NY, const int NZ, const int Lmax, const int kexp, const int what,
                       const float px, const float py, const float pz, const int level, const int cx, const int cy, const int cz,
                       const int obase, SocLtAim &A)
{
    const int  L = level, D = Lmax - L;
    const bool place = (what == SOC_LTM_PLACE), deep = (L > 0);
    const int  om = deep ? ~1 : 0;
    A.ox = cx & om;  A.oy = cy & om;  A.oz = cz & om;
    int Jx, Jy, Jz;                                          // floor(pos * 2^D): the cell within the octet and D digits below it
    {
        const float sc = soc_lt_pow2(D);
        Jx = (int)soc_floorf(px * sc);  Jy = (int)soc_floorf(py * sc);  Jz = (int)soc_floorf(pz * sc);
    }
    if (place) { Jx = (int)((unsigned)(cx - A.ox) << D);  Jy = (int)((unsigned)(cy - A.oy) << D);  Jz = (int)((unsigned)(cz - A.oz) << D); }
    A.Fx = (int)((unsigned)A.ox << D) + Jx;  A.Fy = (int)((unsigned)A.oy << D) + Jy;  A.Fz = (int)((unsigned)A.oz << D) + Jz;
    A.Rx = A.Fx >> Lmax;  A.Ry = A.Fy >> Lmax;  A.Rz = A.Fz >> Lmax;
    // a sibling in the packet's own octet: floor(pos) in {0,1}^3, i.e. nothing of J above bit D
    const bool sib = deep & !place & ((((unsigned)(Jx | Jy | Jz)) >> (D + 1)) == 0u);
    // not for exact geometry to decide (soc_lt_degenerate; a sibling: POS/2 + octant only, bound 2^(kexp+1))
    const float mabs = soc_fminf(soc_fabsf(px), soc_fminf(soc_fabsf(py), soc_fabsf(pz)));
    const float mmin = soc_fminf(px, soc_fminf(py, pz));
    const bool onface = (px == soc_floorf(px)) || (py == soc_floorf(py)) || (pz == soc_floorf(pz));
    const bool slow = deep & !place & (sib ? !(mmin >= soc_lt_pow2(kexp + 1)) : (!(mabs >= soc_lt_pow2(kexp + L)) | onface));
    const bool out0 = !deep & !place & ((px == 0.0f) | (py == 0.0f) | (pz == 0.0f));   // root grid: pos <= 0 leaves the model (:214)
    const int  rx = A.Rx - K.x0, ry = A.Ry - K.y0, rz = A.Rz - K.z0;
    const bool inbox = !(((unsigned)rx >= (unsigned)K.bx) | ((unsigned)ry >= (unsigned)K.by) | ((unsigned)rz >= (unsigned)K.bz));
    const bool outside = out0 | ((unsigned)A.Rx >= (unsigned)NX) | ((unsigned)A.Ry >= (unsigned)NY) | ((unsigned)A.Rz >= (unsigned)NZ);
    const int  sroot = SOC_MAD24(SOC_MAD24(rz, K.by, ry), K.bx, rx);
    const int  ssib  = obase + (((Jx >> D) & 1) | (((Jy >> D) & 1) << 1) | (((Jz >> D) & 1) << 2));
    const bool go = !slow & (sib | (inbox & !out0));
    A.s = go ? (sib ? ssib : sroot) : 0;
    A.l = sib ? L : 0;
    A.r = go ? SOC_LT_INSIDE : (slow ? SOC_LT_SLOW : (place ? SOC_LT_LOST : (outside ? SOC_LT_EXIT : SOC_LT_LEAVE)));
}

// rec = tree[A.s].  Only for A.r == SOC_LT_INSIDE.
template <typename TREE>
SOC_HD int soc_lt_land(const TREE tree, const SocLtAim &A, const int Lmax, const int what, float rec,
                       float &px, float &py, float &pz, int &level, int &cx, int &cy, int &cz, int &slot, int &obase, float &dens)
{
    const int  L = level;
    const bool place = (what == SOC_LTM_PLACE);
    const int  lstop = place ? L : Lmax;
    int s = A.s, l = A.l, base = obase;
    while (!(rec > 0.0f) && (l < lstop)) {
        l++;
        const int sh = Lmax - l;
        base = soc_lt_link(rec);
        s = base + (((A.Fx >> sh) & 1) | (((A.Fy >> sh) & 1) << 1) | (((A.Fz >> sh) & 1) << 2));
        rec = tree[s];
    }
    // (a placement ends on the packet's own level, l == L, with the point F built from its own cell: the lines below then give the
    //  coordinates and the level back as they were -- only the position, which a placement does not touch, is kept by a select)
    const bool lost = place & (l != L);
    const int sh = Lmax - l;
    const int nx = A.Fx >> sh, ny = A.Fy >> sh, nz = A.Fz >> sh;            // the leaf, on its level
    const int qm = (l > 0) ? ~1 : 0;
    // pos' = RN(pos * 2^(l-L) + (O_old * 2^(l-L) - O_new)); the constant is a dyadic number of few bits: exact.  Same level and
    // same octet: pos * 1 + 0 = pos (pos is never a zero here: a coordinate on a cell face went to SOC_LT_SLOW or SOC_LT_EXIT).
    const float sc = soc_lt_pow2(l - L);
    const float qx = SOC_FMA(px, sc, SOC_FMA((float)A.ox, sc, -(float)(nx & qm)));
    const float qy = SOC_FMA(py, sc, SOC_FMA((float)A.oy, sc, -(float)(ny & qm)));
    const float qz = SOC_FMA(pz, sc, SOC_FMA((float)A.oz, sc, -(float)(nz & qm)));
    px = place ? px : qx;  py = place ? py : qy;  pz = place ? pz : qz;
    cx = nx;  cy = ny;  cz = nz;
    level = l;
    slot = s;
    obase = base;
    dens = rec;
    return lost ? SOC_LT_LOST : SOC_LT_INSIDE;
}

// both halves in one call (host harness, slow paths)
template <typename TREE>
SOC_HD int soc_lt_move(const TREE tree, const SocLBrick &K, const int NX, const int NY, const int NZ, const int Lmax, const int kexp,
                       const int what, float &px, float &py, float &pz, int &level, int &cx, int &cy, int &cz, int &slot, int &obase,
                       float &dens, int &Rx, int &Ry, int &Rz)
{
    SocLtAim A;
    soc_lt_aim(K, NX, NY, NZ, Lmax, kexp, what, px, py, pz, level, cx, cy, cz, obase, A);
    Rx = A.Rx;  Ry = A.Ry;  Rz = A.Rz;
    if (A.r != SOC_LT_INSIDE) return A.r;
    return soc_lt_land(tree, A, Lmax, what, tree[A.s], px, py, pz, level, cx, cy, cz, slot, obase, dens);
}

#endif  // SOC_LTREE_H
