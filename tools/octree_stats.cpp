// octree_stats.cpp -- development aid (host only): what a ray's walk through the reference-policy
// octrees of a scene costs, and what share of the records it covers would pass different kinds of
// pre-test.  Builds the trees with the library's own builder, follows rays through them in the
// reference's order (not bit-exact: statistics only) and counts.
//   g++ -O2 -std=c++17 -I raytracer_amd/csrc tools/octree_stats.cpp raytracer_amd/csrc/rt_octree.cpp -o /tmp/octree_stats
//   /tmp/octree_stats <max_depth> mesh0.bin mesh1.bin ... -- ex ey ez ax ay az vfov aspect lx ly lz
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "rt_octree.h"
#include "rt_pretest.h"

using namespace rt;

struct V { double x, y, z; };
static V operator+(V a, V b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static V operator-(V a, V b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V operator*(V a, double s) { return {a.x * s, a.y * s, a.z * s}; }
static double dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V cross(V l, V r) { return {l.y * r.z - r.y * l.z, l.z * r.x - l.x * r.z, l.x * r.y - r.x * l.y}; }
static V norm(V a) { return a * (1 / std::sqrt(dot(a, a))); }

static bool hit_box(const double *lo, const double *hi, V e, V d, double t0, double t1, double &r0, double &r1)
{
    double a[3] = {(lo[0] - e.x) / d.x, (lo[1] - e.y) / d.y, (lo[2] - e.z) / d.z};
    double b[3] = {(hi[0] - e.x) / d.x, (hi[1] - e.y) / d.y, (hi[2] - e.z) / d.z};
    r0 = t0; r1 = t1;
    for (int k = 0; k < 3; k++) { r0 = std::max(r0, std::min(a[k], b[k])); r1 = std::min(r1, std::max(a[k], b[k])); }
    return r0 <= r1;
}
static bool tri_hit(const LeafTri &tr, V e, V d, double t0, double t1, double &tout)
{
    V v0{tr.v0[0], tr.v0[1], tr.v0[2]}, e1{tr.e1[0], tr.e1[1], tr.e1[2]}, e2{tr.e2[0], tr.e2[1], tr.e2[2]};
    V p = cross(d, e2);
    double det = dot(e1, p);
    if (!(std::fabs(det) >= 1e-200)) return false;
    V s = e - v0;
    double u = dot(s, p) / det;
    if (!(u >= 0 && u <= 1)) return false;
    V q = cross(s, e1);
    double v = dot(d, q) / det;
    if (!(v >= 0 && u + v <= 1)) return false;
    double t = dot(e2, q) / det;
    if (!(t >= t0 && t <= t1)) return false;
    tout = t;
    return true;
}

// the bounds of the triangle clipped to the grown node box (as clipped_triangle_box in rt_pretest.h); false if nothing is left
static bool clipped_box(const LeafTri &t, const FlatNode &n, double lo[3], double hi[3])
{
    double ext = 0;
    for (int k = 0; k < 3; k++) ext = std::max(ext, n.hi[k] - n.lo[k]);
    const double grow = ext / 64;
    double poly[2][16][3];
    int cnt = 3, cur = 0;
    for (int k = 0; k < 3; k++) { poly[0][0][k] = t.v0[k]; poly[0][1][k] = t.v0[k] + t.e1[k]; poly[0][2][k] = t.v0[k] + t.e2[k]; }
    for (int k = 0; k < 3 && cnt; k++)
        for (int side = 0; side < 2 && cnt; side++) {
            const double plane = side ? n.hi[k] + grow : n.lo[k] - grow, sgn = side ? -1.0 : 1.0;
            double (*in)[3] = poly[cur], (*out)[3] = poly[cur ^ 1];
            int m = 0;
            for (int i = 0; i < cnt; i++) {
                const double *a = in[i], *b = in[(i + 1) % cnt];
                const double da = sgn * (a[k] - plane), db = sgn * (b[k] - plane);
                if (da >= 0) { for (int c = 0; c < 3; c++) out[m][c] = a[c]; m++; }
                if ((da >= 0) != (db >= 0)) {
                    const double u = da / (da - db);
                    for (int c = 0; c < 3; c++) out[m][c] = a[c] + (b[c] - a[c]) * u;
                    out[m][k] = plane; m++;
                }
            }
            cnt = m; cur ^= 1;
        }
    if (!cnt) return false;
    for (int k = 0; k < 3; k++) lo[k] = hi[k] = poly[cur][0][k];
    for (int i = 1; i < cnt; i++)
        for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], poly[cur][i][k]); hi[k] = std::max(hi[k], poly[cur][i][k]); }
    return true;
}

struct Pre {            // per record
    double qlo[3], qhi[3];   // clipped box in the 8-bit grid, rounded outwards, back in scene units
    bool empty;
};
struct Tree { FlatTree ft; std::vector<Pre> pre; std::vector<int> depth; std::vector<uint32_t> sorted; };   // sorted: per run, record indices in Morton order

static bool ray_box_window(V e, V d, const double *lo, const double *hi, double t0, double t1)
{
    double r0, r1;
    return hit_box(lo, hi, e, d, t0, t1, r0, r1);
}
static uint32_t morton3(uint32_t x, uint32_t y, uint32_t z)
{
    uint32_t m = 0;
    for (int b = 0; b < 8; b++) m |= ((x >> b) & 1u) << (3 * b) | ((y >> b) & 1u) << (3 * b + 1) | ((z >> b) & 1u) << (3 * b + 2);
    return m;
}

struct Stats {
    double visits = 0, recs = 0, cand = 0, hits = 0, runpass = 0;
    double seg[4] = {0, 0, 0, 0}, segpass[4] = {0, 0, 0, 0}, segrecs[4] = {0, 0, 0, 0};   // list order 16, sorted 16, sorted 8, sorted 32
};

int main(int argc, char **argv)
{
    int max_depth = atoi(argv[1]);
    std::vector<Tree> trees;
    int ai = 2;
    for (; ai < argc && strcmp(argv[ai], "--"); ai++) {
        FILE *f = fopen(argv[ai], "rb");
        fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
        std::vector<double> t(sz / 8);
        if (fread(t.data(), 8, t.size(), f) != t.size()) return 1;
        fclose(f);
        trees.emplace_back();
        std::string err;
        if (!build_octree(t.data(), t.size() / 9, 1e-7, 128, 3, max_depth, trees.back().ft, err)) { printf("%s\n", err.c_str()); return 1; }
        Tree &T = trees.back();
        printf("mesh %s: %zu tris, nodes %zu, records %zu\n", argv[ai], t.size() / 9, T.ft.nodes.size(), T.ft.leaf_tris.size());
        T.depth.assign(T.ft.nodes.size(), 0);
        for (size_t i = 0; i < T.ft.nodes.size(); i++)
            if (T.ft.nodes[i].child_base) for (int c = 0; c < 8; c++) T.depth[T.ft.nodes[i].child_base + c] = T.depth[i] + 1;
        T.pre.resize(T.ft.leaf_tris.size());
        for (FlatNode &n : T.ft.nodes) {
            const float scale = node_grid_scale(n);
            for (uint32_t i = n.obj_begin; i < n.obj_begin + n.obj_count; i++) {
                Pre &p = T.pre[i];
                const LeafTri &lt = T.ft.leaf_tris[i];
                double blo[3], bhi[3];
                p.empty = !clipped_box(lt, n, blo, bhi);
                if (p.empty) for (int k = 0; k < 3; k++) { blo[k] = n.lo[k]; bhi[k] = n.lo[k]; }
                for (int k = 0; k < 3; k++) {
                    double glo = (blo[k] - n.lo[k]) * scale + RT_GRID_ORG, ghi = (bhi[k] - n.lo[k]) * scale + RT_GRID_ORG;
                    double flo = std::floor(glo - 2e-3), fhi = std::ceil(ghi + 2e-3);
                    p.qlo[k] = (flo - RT_GRID_ORG) / scale + n.lo[k];
                    p.qhi[k] = (fhi - RT_GRID_ORG) / scale + n.lo[k];
                }
            }
        }
        T.sorted.resize(T.ft.leaf_tris.size());
        for (FlatNode &n : T.ft.nodes) {
            const float scale = node_grid_scale(n);
            std::vector<std::pair<uint32_t, uint32_t>> key(n.obj_count);
            for (uint32_t i = 0; i < n.obj_count; i++) {
                const Pre &p = T.pre[n.obj_begin + i];
                uint32_t c[3];
                for (int k = 0; k < 3; k++) c[k] = (uint32_t)std::min(255.0, std::max(0.0, ((p.qlo[k] + p.qhi[k]) / 2 - n.lo[k]) * scale + RT_GRID_ORG));
                key[i] = {p.empty ? 0xFFFFFFFFu : morton3(c[0], c[1], c[2]), n.obj_begin + i};
            }
            std::sort(key.begin(), key.end());
            for (uint32_t i = 0; i < n.obj_count; i++) T.sorted[n.obj_begin + i] = key[i].second;
        }
    }
    ai++;
    double cv[11];
    for (int k = 0; k < 11; k++) cv[k] = atof(argv[ai + k]);
    V from{cv[0], cv[1], cv[2]}, at{cv[3], cv[4], cv[5]};
    const double vfov = cv[6], aspect = cv[7];
    V light{cv[8], cv[9], cv[10]};
    V w = norm(from - at), u = norm(cross(V{0, 1, 0}, w)), v = cross(w, u);
    const double dist = std::sqrt(dot(at - from, at - from)), hh = 2 * dist * std::tan(vfov * 3.1415926 / 360), ww = hh * aspect;
    V llc = from - u * (ww / 2) - v * (hh / 2) - w * dist;

    const int MAXD = 16;
    std::vector<Stats> byDepth(MAXD + 1);
    std::vector<Stats> byLen(12);     // run length buckets: log2
    double nrays = 0, ninner = 0, nleaf = 0, nempty = 0, nempty_nohit = 0, ninner_noobj = 0;
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> U(0, 1);

    auto trace = [&](V e, V d, double t0, double t1, int &hm, uint32_t &hrec) -> double {
        nrays++;
        hm = -1;
        for (size_t mi = 0; mi < trees.size(); mi++) {
            Tree &T = trees[mi];
            const FlatTree &ft = T.ft;
            double r0, r1;
            if (!hit_box(ft.nodes[0].lo, ft.nodes[0].hi, e, d, t0, t1, r0, r1)) continue;
            double mint = t1 + 1; uint32_t minrec = ~0u;
            struct Ent { uint32_t node; double tin; };
            std::vector<Ent> stack{{0, r0}};
            while (!stack.empty()) {
                Ent en = stack.back(); stack.pop_back();
                if (en.node != 0 && mint < en.tin) break;
                const FlatNode &n = ft.nodes[en.node];
                double n0, n1;
                if (!hit_box(n.lo, n.hi, e, d, t0, t1, n0, n1)) continue;
                if (n.child_base) ninner++; else nleaf++;
                if (!n.child_base && !n.obj_count) { nempty++; if (minrec == ~0u) nempty_nohit++; }
                if (n.child_base && !n.obj_count) ninner_noobj++;
                bool improved = false;
                if (n.obj_count) {
                    Stats &S = byDepth[std::min(T.depth[en.node], MAXD)];
                    int lb = 0; while ((1u << (lb + 1)) <= n.obj_count && lb < 11) lb++;
                    Stats &L = byLen[lb];
                    S.visits++; L.visits++;
                    double tempt = n1; uint32_t best = ~0u;
                    {
                        double glo[3] = {1e300, 1e300, 1e300}, ghi[3] = {-1e300, -1e300, -1e300};
                        for (uint32_t i = 0; i < n.obj_count; i++) {
                            const Pre &p = T.pre[n.obj_begin + i];
                            if (p.empty) continue;
                            for (int k = 0; k < 3; k++) { glo[k] = std::min(glo[k], p.qlo[k]); ghi[k] = std::max(ghi[k], p.qhi[k]); }
                        }
                        if (ray_box_window(e, d, glo, ghi, n0, n1)) { S.runpass++; L.runpass++; }
                    }
                    for (int variant = 0; variant < 4; variant++) {
                        const uint32_t gs = variant == 2 ? 8 : (variant == 3 ? 32 : 16);
                        for (uint32_t g0 = 0; g0 < n.obj_count; g0 += gs) {
                            double glo[3] = {1e300, 1e300, 1e300}, ghi[3] = {-1e300, -1e300, -1e300};
                            const uint32_t ge = std::min(g0 + gs, n.obj_count);
                            bool any = false;
                            for (uint32_t i = g0; i < ge; i++) {
                                const Pre &p = T.pre[variant ? T.sorted[n.obj_begin + i] : n.obj_begin + i];
                                if (p.empty) continue;
                                any = true;
                                for (int k = 0; k < 3; k++) { glo[k] = std::min(glo[k], p.qlo[k]); ghi[k] = std::max(ghi[k], p.qhi[k]); }
                            }
                            const bool pass = any && ray_box_window(e, d, glo, ghi, n0, n1);
                            S.seg[variant]++; L.seg[variant]++;
                            if (pass) { S.segpass[variant]++; L.segpass[variant]++; S.segrecs[variant] += ge - g0; L.segrecs[variant] += ge - g0; }
                        }
                    }
                    for (uint32_t i = n.obj_begin; i < n.obj_begin + n.obj_count; i++) {
                        const Pre &p = T.pre[i];
                        S.recs++; L.recs++;
                        if (!p.empty && ray_box_window(e, d, p.qlo, p.qhi, n0, n1)) { S.cand++; L.cand++; }
                        double tt;
                        if (tri_hit(ft.leaf_tris[i], e, d, n0, tempt, tt)) { tempt = tt; best = i; S.hits++; L.hits++; }
                    }
                    if (best != ~0u && tempt < mint) { mint = tempt; minrec = best; improved = true; }
                }
                if (!n.child_base) { if (improved) break; continue; }
                Ent ch[8]; int nc = 0;
                for (int c = 0; c < 8; c++) {
                    const FlatNode &cn = ft.nodes[n.child_base + c];
                    double c0, c1;
                    if (hit_box(cn.lo, cn.hi, e, d, t0, t1, c0, c1)) ch[nc++] = {n.child_base + (uint32_t)c, c0};
                }
                std::stable_sort(ch, ch + nc, [](const Ent &a, const Ent &b) { return a.tin < b.tin; });
                for (int c = nc - 1; c >= 0; c--) stack.push_back(ch[c]);
            }
            if (minrec != ~0u && mint <= t1) { t1 = mint; hm = (int)mi; hrec = minrec; }
        }
        return hm >= 0 ? t1 : -1;
    };

    const int GW = 160, GH = 90;
    for (int y = 0; y < GH; y++)
        for (int x = 0; x < GW; x++) {
            V d = norm(llc + u * ((x + U(rng)) / GW * ww) + v * ((y + U(rng)) / GH * hh) - from);
            V e = from;
            for (int b = 0; b < 5; b++) {
                int hm; uint32_t hrec;
                double t = trace(e, d, b ? 1e-9 : 0, 1e10, hm, hrec);
                if (t < 0) break;
                V p = e + d * t;
                const LeafTri &lt = trees[hm].ft.leaf_tris[hrec];
                V nrm = norm(cross(V{lt.e1[0], lt.e1[1], lt.e1[2]}, V{lt.e2[0], lt.e2[1], lt.e2[2]}));
                // shadow ray towards a point of the light
                V lp = light + V{(U(rng) - 0.5) * 120, 0, (U(rng) - 0.5) * 40};
                V l = lp - p; double ld = std::sqrt(dot(l, l)); l = l * (1 / ld);
                if (dot(nrm, l) > 0) { int a; uint32_t r; trace(p, l, 1e-9, ld - 1e-9, a, r); }
                // cosine bounce about the raw normal
                double r1 = 2 * 3.1415926 * U(rng), r2 = U(rng);
                V wv = nrm, uv = norm(cross(std::fabs(wv.x) > 0.1 ? V{0, 1, 0} : V{1, 0, 0}, wv)), vv = cross(wv, uv);
                d = norm(uv * (std::cos(r1) * std::sqrt(r2)) + vv * (std::sin(r1) * std::sqrt(r2)) + wv * std::sqrt(1 - r2));
                e = p;
            }
        }
    printf("rays %.0f  inner/ray %.2f  leaf/ray %.2f  empty leaves/ray %.2f (before any hit: %.2f)  inner without objects/ray %.2f\n", nrays, ninner / nrays, nleaf / nrays, nempty / nrays, nempty_nohit / nrays, ninner_noobj / nrays);
    auto row = [&](const char *name, int k, const Stats &S) {
        if (!S.visits) return;
        printf("%s %2d: visits/ray %6.3f run-box pass %5.1f%% recs/ray %7.2f cand %6.2f hit %5.2f |", name, k, S.visits / nrays, 100 * S.runpass / S.visits, S.recs / nrays, S.cand / nrays, S.hits / nrays);
        const char *vn[4] = {"list16", "sort16", "sort8", "sort32"};
        for (int v = 0; v < 4; v++) printf(" %s: segs %6.2f pass %5.1f%% recs-after %6.2f |", vn[v], S.seg[v] / nrays, 100 * S.segpass[v] / std::max(S.seg[v], 1.0), S.segrecs[v] / nrays);
        printf("\n");
    };
    Stats tot;
    for (int k = 0; k <= MAXD; k++) {
        row("depth", k, byDepth[k]);
        const Stats &S = byDepth[k];
        tot.visits += S.visits; tot.recs += S.recs; tot.cand += S.cand; tot.hits += S.hits; tot.runpass += S.runpass;
        for (int v = 0; v < 4; v++) { tot.seg[v] += S.seg[v]; tot.segpass[v] += S.segpass[v]; tot.segrecs[v] += S.segrecs[v]; }
    }
    row("total", 0, tot);
    for (int k = 0; k < 12; k++) row("len2^", k, byLen[k]);
    return 0;
}
