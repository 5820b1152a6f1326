// Micro-benchmark (development aid): rate at which 64 lanes that each walk their own run of
// consecutive 80-byte records can fetch them, (A) every lane loading its record with five 16-byte
// loads, (B) the four lanes of a quad loading the first 64 bytes of each other's records together
// (one contiguous 64-byte access per record) plus one 16-byte tail load each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at line %d\n", (int)e_, __LINE__); return 1; } } while (0)

__global__ void k(const double2 *data, int runs, int runlen, int mode, unsigned nrec, double *out)
{
    const unsigned lane = threadIdx.x & 63u;
    double acc = 0;
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x, quad0 = tid & ~3u;
    unsigned sd[4];
    for (int j = 0; j < 4; j++) sd[j] = (quad0 + j) * 2654435761u + 12345u;      // the seeds of the quad's four lanes
    for (int r = 0; r < runs; r++) {
        unsigned st[4];
        for (int j = 0; j < 4; j++) { sd[j] = sd[j] * 1664525u + 1013904223u; st[j] = (sd[j] >> 8) % (nrec - runlen); }
        const unsigned start = st[lane & 3u];
        for (int i = 0; i < runlen; i++) {
            const size_t rec = start + i;
            if (mode == 0) {
                const double2 *p = data + rec * 5;
                double2 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
                acc += a.x + b.y + c.x + d.y + e.x;
            } else if (mode == 1) {
                // chunk (lane & 3) of the records of the quad's four lanes, then the own tail
                const double2 *base = data + (lane & 3u);
                double2 v[4];
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = base[(size_t)(st[j] + i) * 5];
                double2 e = data[rec * 5 + 4];
                acc += v[0].x + v[1].y + v[2].x + v[3].y + e.x;
            } else {
                // as mode 1 with 128-byte records (the 64-byte block is aligned)
                const double2 *base = data + (lane & 3u);
                double2 v[4];
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = base[(size_t)((st[j] + i) % (nrec * 5 / 8)) * 8];
                double2 e = data[(rec % (nrec * 5 / 8)) * 8 + 4];
                acc += v[0].x + v[1].y + v[2].x + v[3].y + e.x;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main()
{
    const unsigned nrec = 110000;                       // 8.8 MB of 80-byte records
    std::vector<double2> h((size_t)nrec * 5);
    for (auto &x : h) { x.x = 1.0; x.y = 2.0; }
    double2 *d; double *out;
    CHECK(hipMalloc(&d, h.size() * sizeof(double2)));
    CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(double2), hipMemcpyHostToDevice));
    const int blocks = 256 * 3, threads = 256, runs = 40, runlen = 32;
    CHECK(hipMalloc(&out, (size_t)blocks * threads * sizeof(double)));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int mode = 0; mode < 3; mode++) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 2, runlen, mode, nrec, out);
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, runs, runlen, mode, nrec, out);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        double recs = (double)blocks * threads * runs * runlen;
        printf("mode %d: %.3f ms, %.1f G records/s\n", mode, ms, recs / ms / 1e6);
    }
    return 0;
}
