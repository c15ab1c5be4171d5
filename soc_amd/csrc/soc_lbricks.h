// soc_lbricks.h -- host-side construction of the brick-local hierarchies of soc_ltree.h.  Plain C++ (no HIP), so the
// same code is compiled into libsoc_hip.so and -- with -fsanitize=address,undefined -- into the CPU tests.
//
// A brick is a box of root cells together with everything below them, at most `cap` cells (leaves and refined
// cells).  The root grid is cut into tiles of 16^3 cells; a tile that holds more than `cap` cells is halved along
// its longest edge until every part fits.  Within a brick the cells get slots: the root cells of the box first
// (x fastest), then the octets of refined cells in breadth-first order, eight consecutive slots per octet; the
// entry of a refined cell is the link to the first slot of its octet in the cloud file's encoding
// (-(float with the bits of the index)), the entry of a leaf is its density.
#ifndef SOC_LBRICKS_H
#define SOC_LBRICKS_H

#include <cstdint>
#include <cstring>
#include <vector>

#include "soc_ltree.h"

struct SocLBricksHost {
    std::vector<SocLBrick> bricks;
    std::vector<float>     btree;      // [cells with a slot] density or local link
    std::vector<int>       bcell;      // [cells with a slot] global cell index
    std::vector<int>       rbrick;     // [NX*NY*NZ] brick of every root cell
    int  max_slots = 0;
    bool ok = false;                   // false: some root cell holds more than cap cells (or a link is broken)
};

namespace soc_lb {

inline int link_of(float d) { float m = -d;  int i;  memcpy(&i, &m, 4);  return i; }
inline float link_to(int i) { float m;  memcpy(&m, &i, 4);  return -m; }

struct Builder {
    int NX, NY, NZ, LEVELS, cap;
    const int *LCELLS, *OFF;
    const float *D;
    std::vector<uint32_t> sub;         // cells in the subtree of every root cell (saturating)
    SocLBricksHost &out;

    Builder(int nx, int ny, int nz, int levels, const int *lcells, const int *off, const float *dens, int cap_, SocLBricksHost &o)
        : NX(nx), NY(ny), NZ(nz), LEVELS(levels), cap(cap_), LCELLS(lcells), OFF(off), D(dens), out(o) {}

    bool count()
    {
        // cells per subtree, bottom-up over the levels; only the root level is kept
        std::vector<uint32_t> below, cur;
        for (int l = LEVELS - 1; l >= 0; l--) {
            cur.assign((size_t)LCELLS[l], 1u);
            const float *d = D + OFF[l];
            for (int i = 0; i < LCELLS[l]; i++) {
                if (d[i] > 0.0f) continue;
                if (l + 1 >= LEVELS) return false;                   // a link on the last level
                const int c = link_of(d[i]);
                if (c < 0 || c + 8 > LCELLS[l + 1]) return false;
                uint64_t n = 1;
                for (int k = 0; k < 8; k++) n += below[(size_t)c + k];
                cur[i] = (uint32_t)(n > 0x7fffffffu ? 0x7fffffffu : n);
            }
            below.swap(cur);
        }
        sub.swap(below);
        return true;
    }
    uint64_t cells_in(int x0, int y0, int z0, int dx, int dy, int dz) const
    {
        uint64_t n = 0;
        for (int z = z0; z < z0 + dz; z++)
            for (int y = y0; y < y0 + dy; y++) {
                const uint32_t *row = sub.data() + ((size_t)z * NY + y) * NX;
                for (int x = x0; x < x0 + dx; x++) n += row[x];
            }
        return n;
    }
    bool emit(int x0, int y0, int z0, int dx, int dy, int dz, int n)
    {
        SocLBrick K;
        K.x0 = x0;  K.y0 = y0;  K.z0 = z0;  K.bx = dx;  K.by = dy;  K.bz = dz;
        K.base = (int)out.btree.size();
        K.nslot = n;
        const int id = (int)out.bricks.size();
        const size_t base = out.btree.size();
        out.btree.resize(base + n);
        out.bcell.resize(base + n);
        // (global cell, level) of every slot; refined cells are expanded in slot order: breadth first
        std::vector<int> lev((size_t)n);
        int fill = 0;
        for (int z = z0; z < z0 + dz; z++)
            for (int y = y0; y < y0 + dy; y++)
                for (int x = x0; x < x0 + dx; x++) {
                    const int r = (z * NY + y) * NX + x;
                    out.rbrick[r] = id;
                    out.bcell[base + fill] = r;
                    lev[fill] = 0;
                    fill++;
                }
        for (int s = 0; s < fill; s++) {
            const int g = out.bcell[base + s];
            const float d = D[g];
            if (d > 0.0f) { out.btree[base + s] = d;  continue; }
            if (fill + 8 > n) return false;
            const int l = lev[s], c = OFF[l + 1] + link_of(d);
            out.btree[base + s] = link_to(fill);
            for (int k = 0; k < 8; k++) { out.bcell[base + fill] = c + k;  lev[fill] = l + 1;  fill++; }
        }
        if (fill != n) return false;
        out.bricks.push_back(K);
        if (n > out.max_slots) out.max_slots = n;
        return true;
    }
    bool split(int x0, int y0, int z0, int dx, int dy, int dz)
    {
        const uint64_t n = cells_in(x0, y0, z0, dx, dy, dz);
        if (n <= (uint64_t)cap) return emit(x0, y0, z0, dx, dy, dz, (int)n);
        if (dx == 1 && dy == 1 && dz == 1) return false;             // one root cell with more than cap cells below it
        if (dz >= dy && dz >= dx) { const int h = (dz + 1) / 2;  return split(x0, y0, z0, dx, dy, h) && split(x0, y0, z0 + h, dx, dy, dz - h); }
        if (dy >= dx)             { const int h = (dy + 1) / 2;  return split(x0, y0, z0, dx, h, dz) && split(x0, y0 + h, z0, dx, dy - h, dz); }
        const int h = (dx + 1) / 2;
        return split(x0, y0, z0, h, dy, dz) && split(x0 + h, y0, z0, dx - h, dy, dz);
    }
    bool build()
    {
        out = SocLBricksHost();
        if (!count()) return false;
        out.rbrick.assign((size_t)NX * NY * NZ, -1);
        size_t total = 0;
        for (int l = 0; l < LEVELS; l++) total += (size_t)LCELLS[l];
        out.btree.reserve(total);
        out.bcell.reserve(total);
        const int T = 16;
        for (int z = 0; z < NZ; z += T)
            for (int y = 0; y < NY; y += T)
                for (int x = 0; x < NX; x += T) {
                    const int dx = (x + T <= NX) ? T : NX - x, dy = (y + T <= NY) ? T : NY - y, dz = (z + T <= NZ) ? T : NZ - z;
                    if (!split(x, y, z, dx, dy, dz)) return false;
                }
        out.ok = true;
        return true;
    }
};

}  // namespace soc_lb

// Returns false (and out.ok == false) when the hierarchy cannot be cut into such bricks: the caller keeps the
// sweep that reads the hierarchy from global memory.
inline bool soc_lbricks_build(int NX, int NY, int NZ, int LEVELS, const int *LCELLS, const int *OFF, const float *DENS, int cap,
                              SocLBricksHost &out)
{
    soc_lb::Builder B(NX, NY, NZ, LEVELS, LCELLS, OFF, DENS, cap, out);
    const bool ok = B.build();
    if (!ok) out = SocLBricksHost();
    return ok;
}

#endif  // SOC_LBRICKS_H
