// soc_octbricks.h -- host side of the brick sweep on hierarchies that stay in global memory: cut the hierarchy into
// bricks of <= CAP cells (leaves and the refined cells above them) and give every cell its (brick, slot) word.
// Plain C++ (no HIP): soc_brick.hip uploads the result; tests/ltree_host.cpp runs it under ASan/UBSan.
#ifndef SOC_OCTBRICKS_H
#define SOC_OCTBRICKS_H

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#define SOC_SLOT_BITS 14
#define SOC_SLOT_MASK ((1u << SOC_SLOT_BITS) - 1u)

struct SocOctBuilder {
    const int NX, NY, NZ, LEVELS, CELLS;
    const int *LCELLS, *OFF;
    const float *D;
    std::vector<uint32_t> sub;                 // cells (leaves + refined cells) in the subtree of every cell
    std::vector<uint32_t> slotmap;             // per cell: brick << SOC_SLOT_BITS | slot
    std::vector<int> bcell, bbase;             // cells in brick order; first entry of every brick (+ end)
    int CAP, fill = 0;

    SocOctBuilder(int nx, int ny, int nz, int levels, int cells, const int *lcells, const int *off, const float *d, int cap)
        : NX(nx), NY(ny), NZ(nz), LEVELS(levels), CELLS(cells), LCELLS(lcells), OFF(off), D(d), CAP(cap) {}

    static int link(float d) { float m = -d;  int i;  memcpy(&i, &m, 4);  return i; }

    void count()
    {
        sub.assign((size_t)CELLS, 0u);
        for (int l = LEVELS - 1; l >= 0; l--) {
            const size_t o = (size_t)OFF[l];
            for (int i = 0; i < LCELLS[l]; i++) {
                const float d = D[o + i];
                if (d > 0.0f) { sub[o + i] = 1;  continue; }
                uint32_t n = 1;                              // the refined cell itself holds a slot too (see place_subtree)
                if (l + 1 < LEVELS) {
                    const size_t c = (size_t)OFF[l + 1] + link(d);
                    for (int k = 0; k < 8; k++) n += sub[c + k];
                }
                sub[o + i] = n;
            }
        }
    }
    void open(uint32_t need)
    {
        if (fill > 0 && fill + (long long)need > CAP) { bbase.push_back((int)bcell.size());  fill = 0; }
    }
    void place_subtree(int l, int i)            // all leaves below (l, i) into the open brick
    {
        const size_t a = (size_t)OFF[l] + i;
        if (D[a] > 0.0f) {
            slotmap[a] = ((uint32_t)(bbase.size() - 1) << SOC_SLOT_BITS) | (uint32_t)fill;
            bcell.push_back((int)a);
            fill++;
            return;
        }
        // A refined cell takes no part in the transfer -- except that SimRAM_CL without emission weights sends
        // packets from EVERY cell index (kernel_ASOC.c:1318-1355 has no leaf test there): such a packet starts "in"
        // the refined cell, with the link as its density, and its first step is tallied there.  So it has a slot.
        place_cell(a);
        if (l + 1 >= LEVELS) return;
        const int c = link(D[a]);
        for (int k = 0; k < 8; k++) place_subtree(l + 1, c + k);
    }
    void place_cell(size_t a)
    {
        slotmap[a] = ((uint32_t)(bbase.size() - 1) << SOC_SLOT_BITS) | (uint32_t)fill;
        bcell.push_back((int)a);
        fill++;
    }
    void assign_subtree(int l, int i)
    {
        const uint32_t n = sub[(size_t)OFF[l] + i];
        if (n == 0) return;
        if (n <= (uint32_t)CAP) { open(n);  place_subtree(l, i);  return; }
        open(1);
        place_cell((size_t)OFF[l] + i);                    // the refined cell itself, then its children one by one
        const int c = link(D[(size_t)OFF[l] + i]);
        for (int k = 0; k < 8; k++) assign_subtree(l + 1, c + k);
    }
    unsigned long long count_cube(int x0, int y0, int z0, int s) const
    {
        unsigned long long n = 0;
        for (int z = z0; z < std::min(z0 + s, NZ); z++)
            for (int y = y0; y < std::min(y0 + s, NY); y++)
                for (int x = x0; x < std::min(x0 + s, NX); x++) n += sub[((size_t)z * NY + y) * NX + x];
        return n;
    }
    void assign_cube(int x0, int y0, int z0, int s)
    {
        if (x0 >= NX || y0 >= NY || z0 >= NZ) return;
        const unsigned long long n = count_cube(x0, y0, z0, s);
        if (n == 0) return;
        if (n <= (unsigned long long)CAP) {
            open((uint32_t)n);
            for (int z = z0; z < std::min(z0 + s, NZ); z++)
                for (int y = y0; y < std::min(y0 + s, NY); y++)
                    for (int x = x0; x < std::min(x0 + s, NX); x++) place_subtree(0, (z * NY + y) * NX + x);
            return;
        }
        if (s == 1) { assign_subtree(0, (z0 * NY + y0) * NX + x0);  return; }
        const int h = s / 2;
        for (int k = 0; k < 8; k++) assign_cube(x0 + (k & 1) * h, y0 + ((k >> 1) & 1) * h, z0 + (k >> 2) * h, h);
    }
    void build()
    {
        count();
        slotmap.assign((size_t)CELLS, 0xffffffffu);
        bcell.clear();
        bbase.assign(1, 0);
        fill = 0;
        for (int z = 0; z < NZ; z += 16)
            for (int y = 0; y < NY; y += 16)
                for (int x = 0; x < NX; x += 16) assign_cube(x, y, z, 16);
        if (fill > 0) bbase.push_back((int)bcell.size());
    }
};

#endif  // SOC_OCTBRICKS_H
