// rt_pretest.h -- host side of the kernels' single-precision pre-test: the bounds that are stored beside
// the records of a node's run, formed once at upload.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#include "rt_device.h"

namespace rt {

// Units per length of a node's grid, as the float the kernels multiply with (it is stored in the
// node record; the kernels use it to bring the ray into the same grid); 0 (pre-test off for
// the node: every record gets the whole-node box) when the box is degenerate or so large or
// small that single precision cannot hold the scale.  The lowest mantissa bit is cleared because
// the node record keeps a flag there.
inline float node_grid_scale(const FlatNode &n)
{
    double ext = 0;
    for (int k = 0; k < 3; k++) {
        const double e = n.hi[k] - n.lo[k];
        if (!std::isfinite(e) || !std::isfinite(n.lo[k])) return 0.0f;
        if (e > ext) ext = e;
    }
    if (!(ext > 0)) return 0.0f;
    const double sc = (double)RT_GRID_EDGE / ext;
    if (!(sc > 1e-30 && sc < 1e30)) return 0.0f;
    float f = (float)sc;
    uint32_t bits;
    std::memcpy(&bits, &f, 4);
    bits &= ~1u;                                      // rounds the scale down by at most one ulp: the box still fits
    if ((double)f > sc) bits -= 2u;                   // never above RT_GRID_EDGE / ext
    std::memcpy(&f, &bits, 4);
    return f;
}

// ---------------------------------------------------------------- pre-test boxes
// The kernels' single-precision pre-test asks whether the ray crosses a box around the part of the
// record's primitive that lies inside the node's box, within the node's window.  Boxes are stored
// as bytes in the node's own grid (the node's box starts at RT_GRID_ORG on every axis, its longest
// edge is RT_GRID_EDGE units): {lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, 0, 0}, rounded outwards.
// A primitive test only accepts a hit inside the node's window, the stretch of the ray inside the
// node's box (octree.cpp:19-38 with the window of algorithms.h:223-236), so the part of a triangle
// outside the box cannot be hit from this run.  The node's box is grown by 1/64 of its longest edge
// on every side first: the window's ends are rounded distances, and the hit point of a distance
// inside the window may lie outside the box by that rounding.
struct BoxD { double lo[3], hi[3]; bool ok, empty; };

// bounds of the triangle (v0, v0+e1, v0+e2) clipped against the node's grown box
// (Sutherland-Hodgman against the six planes); ok = false: could not be formed
inline BoxD clipped_triangle_box(const LeafTri &t, const FlatNode &n)
{
    BoxD b;
    b.ok = false; b.empty = false;
    double ext = 0;
    for (int k = 0; k < 3; k++) { const double e = n.hi[k] - n.lo[k]; if (!(e <= ext)) ext = e; b.lo[k] = n.lo[k]; b.hi[k] = n.hi[k]; }
    if (!std::isfinite(ext) || !(ext > 0)) return b;
    // A triangle whose corners lie (almost) on one line has no box: the double-precision test divides by a
    // determinant that is rounding noise for it, so what it accepts need not lie near the triangle at all
    // (corners on a lattice make exactly collinear triangles, and the test "hits" them at distance 0 from
    // anywhere); with the whole-node box every ray that visits the node gets that test, as in the reference walk.
    {
        const double cx = t.e1[1] * t.e2[2] - t.e2[1] * t.e1[2], cy = t.e1[2] * t.e2[0] - t.e1[0] * t.e2[2], cz = t.e1[0] * t.e2[1] - t.e2[0] * t.e1[1];
        const double l1 = t.e1[0] * t.e1[0] + t.e1[1] * t.e1[1] + t.e1[2] * t.e1[2], l2 = t.e2[0] * t.e2[0] + t.e2[1] * t.e2[1] + t.e2[2] * t.e2[2];
        if (!(cx * cx + cy * cy + cz * cz > 1e-12 * l1 * l2)) return b;       // sine of the angle at v0 under 1e-6 (or not a number)
    }
    const double grow = ext / 64;
    double poly[2][16][3];
    int cnt = 3, cur = 0;
    for (int k = 0; k < 3; k++) { poly[0][0][k] = t.v0[k]; poly[0][1][k] = t.v0[k] + t.e1[k]; poly[0][2][k] = t.v0[k] + t.e2[k]; }
    for (int k = 0; k < 3 && cnt; k++) {
        for (int side = 0; side < 2 && cnt; side++) {
            const double plane = side ? n.hi[k] + grow : n.lo[k] - grow;
            const double sgn = side ? -1.0 : 1.0;                 // inside: sgn * (x - plane) >= 0
            double (*in)[3] = poly[cur], (*out)[3] = poly[cur ^ 1];
            int m = 0;
            for (int i = 0; i < cnt; i++) {
                const double *a = in[i], *c = in[(i + 1) % cnt];
                const double da = sgn * (a[k] - plane), db = sgn * (c[k] - plane);
                if (!std::isfinite(da) || !std::isfinite(db)) return b;
                if (da >= 0) { for (int q = 0; q < 3; q++) out[m][q] = a[q]; m++; }
                if ((da >= 0) != (db >= 0)) {
                    const double u = da / (da - db);
                    for (int q = 0; q < 3; q++) out[m][q] = a[q] + (c[q] - a[q]) * u;
                    out[m][k] = plane;
                    m++;
                }
                if (m > 14) return b;                             // cannot happen: a triangle cut by six planes has at most nine corners
            }
            cnt = m; cur ^= 1;
        }
    }
    b.ok = true;
    if (cnt == 0) { b.empty = true; return b; }
    for (int k = 0; k < 3; k++) { b.lo[k] = b.hi[k] = poly[cur][0][k]; }
    for (int i = 1; i < cnt; i++)
        for (int k = 0; k < 3; k++) { if (poly[cur][i][k] < b.lo[k]) b.lo[k] = poly[cur][i][k]; if (poly[cur][i][k] > b.hi[k]) b.hi[k] = poly[cur][i][k]; }
    for (int k = 0; k < 3; k++) if (!std::isfinite(b.lo[k]) || !std::isfinite(b.hi[k])) { b.ok = false; return b; }
    return b;
}
// a voxel (v0 = lower corner, e1 = upper corner) cut to the node's grown box
inline BoxD clipped_voxel_box(const LeafTri &t, const FlatNode &n)
{
    BoxD b;
    b.ok = false; b.empty = false;
    double ext = 0;
    for (int k = 0; k < 3; k++) { const double e = n.hi[k] - n.lo[k]; if (!(e <= ext)) ext = e; b.lo[k] = n.lo[k]; b.hi[k] = n.hi[k]; }
    if (!std::isfinite(ext) || !(ext > 0)) return b;
    const double grow = ext / 64;
    for (int k = 0; k < 3; k++) {
        if (!std::isfinite(t.v0[k]) || !std::isfinite(t.e1[k])) return b;
        b.lo[k] = std::fmax(t.v0[k], n.lo[k] - grow);
        b.hi[k] = std::fmin(t.e1[k], n.hi[k] + grow);
        if (b.lo[k] > b.hi[k]) b.empty = true;
    }
    b.ok = true;
    return b;
}
// the whole grid (0 .. 255 on every axis): every ray that crosses the node passes.  A valid stand-in for
// any record whose own box cannot be expressed: the primitive tests only accept hits inside the node's window.
inline void store_whole_node_box(uint32_t *dst) { dst[0] = 0xFF000000u; dst[1] = 0x0000FFFFu; }
// An empty box: the single grid point (0, 0, 0), at least RT_GRID_ORG - 3 units outside the node's grown box
// on every axis, so no ray reaches it inside the node's window.
inline void store_empty_box(uint32_t *dst) { dst[0] = 0u; dst[1] = 0u; }
// The box in the node's grid, rounded outwards: RT_BOX_GROW grid units beyond the bounds (the kernels'
// single-precision arithmetic places the ray within 2e-4 units; the rest is for the error of the
// double-precision primitive test, see rt_device.h), then on to the next integer.
// grid_lo / grid_hi (nullable) receive the stored integers, for the union over a segment.
inline void pack_box(uint32_t *dst, const BoxD &b, const FlatNode &n, float scale, int *grid_lo = nullptr, int *grid_hi = nullptr)
{
    store_whole_node_box(dst);
    if (grid_lo) for (int k = 0; k < 3; k++) { grid_lo[k] = 0; grid_hi[k] = 255; }
    if (!b.ok || !(scale > 0.0f)) return;
    if (b.empty) {
        store_empty_box(dst);
        if (grid_lo) for (int k = 0; k < 3; k++) { grid_lo[k] = 256; grid_hi[k] = -1; }     // nothing to add to a union
        return;
    }
    const double sc = (double)scale;
    uint32_t lo[3], hi[3];
    for (int k = 0; k < 3; k++) {
        const double glo = (b.lo[k] - n.lo[k]) * sc + (double)RT_GRID_ORG, ghi = (b.hi[k] - n.lo[k]) * sc + (double)RT_GRID_ORG;
        if (!std::isfinite(glo) || !std::isfinite(ghi)) return;
        double l = std::floor(glo - RT_BOX_GROW), h = std::ceil(ghi + RT_BOX_GROW);
        if (l < 0) l = 0;
        if (h > 255) h = 255;
        if (l > 255 || h < 0) return;
        lo[k] = (uint32_t)l; hi[k] = (uint32_t)h;
    }
    dst[0] = lo[0] | (lo[1] << 8) | (lo[2] << 16) | (hi[0] << 24);
    dst[1] = hi[1] | (hi[2] << 8);
    if (grid_lo) for (int k = 0; k < 3; k++) { grid_lo[k] = (int)lo[k]; grid_hi[k] = (int)hi[k]; }
}
inline void store_grid_box(uint32_t *dst, const int *lo, const int *hi)
{
    if (lo[0] > hi[0] || lo[1] > hi[1] || lo[2] > hi[2]) { store_empty_box(dst); return; }
    dst[0] = (uint32_t)lo[0] | ((uint32_t)lo[1] << 8) | ((uint32_t)lo[2] << 16) | ((uint32_t)hi[0] << 24);
    dst[1] = (uint32_t)hi[1] | ((uint32_t)hi[2] << 8);
}

} // namespace rt
