// rt_multi.cpp -- more than one GPU behind the C ABI (declared in include/rt_hip.h):
//   * rt_comm_* / rt_reduce_and_resolve: the one exchange step of a frame whose samples are
//     partitioned across devices -- the devices' un-normalised radiance sums are added onto the
//     root with one RCCL reduce (fp64, in place), and the root resolves the frame to bytes.  The
//     clamp of Scene::run comes after the average (scene.cpp:112-113), so sums must meet unclamped.
//   * rt_multi_*: Scene::runParallelly (scene.cpp:141-172) for one process that drives several
//     devices: one host thread and one scene handle per device (the reference clones its scene per
//     thread, scene.cpp:150), each thread renders its slice of the sample indices, then the
//     reduce above.
// RCCL is loaded at first use (dlopen), so the library has no link-time dependency on it and a
// process that already holds an RCCL (torch's) keeps using that one.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_internal.h"

namespace {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};

// The process's RCCL, or null with the reason in `why`.
const Rccl *rccl(std::string &why)
{
    static std::once_flag once;
    static Rccl R;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so"};
        for (const char *n : names) if (!R.lib) R.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);     // one that is loaded already
        for (const char *n : names) if (!R.lib) R.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!R.lib) {
            const char *why = dlerror();                  // a second call would return null: the first one clears it
            R.err = std::string("RCCL is not loadable: ") + (why ? why : "librccl.so.1 not found");
            return;
        }
        auto sym = [&](const char *name) { void *p = dlsym(R.lib, name); if (!p && R.err.empty()) R.err = std::string("RCCL lacks ") + name; return p; };
        R.GetUniqueId = (decltype(R.GetUniqueId))sym("ncclGetUniqueId");
        R.CommInitRank = (decltype(R.CommInitRank))sym("ncclCommInitRank");
        R.CommInitAll = (decltype(R.CommInitAll))sym("ncclCommInitAll");
        R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
        R.CommAbort = (decltype(R.CommAbort))sym("ncclCommAbort");
        R.Reduce = (decltype(R.Reduce))sym("ncclReduce");
        R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
    });
    if (!R.err.empty()) { why = R.err; return nullptr; }
    return &R;
}

int fail_nccl(const Rccl *R, const char *what, ncclResult_t e)
{
    return rt::set_error(std::string(what) + ": " + (R && R->GetErrorString ? R->GetErrorString(e) : "RCCL error"), -3);
}

} // namespace

// A communicator: `n` local members (devices of this process), each with its RCCL handle; rank of
// local member i = rank0 + i; the root (rank 0) is local member 0 of the process that holds it.
struct rt_comm {
    int nranks = 0, rank0 = 0;
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
    std::mutex mu;
    bool aborted = false;       // a member could not enter a collective: the handles were aborted and are gone
};

namespace {

// A local member cannot enter the collective its peers are entering (or already wait in): abort
// every local handle, which makes the collective's kernels on the peers' streams give up instead
// of waiting for ever.  The communicator is unusable afterwards.
void abort_comm(rt_comm *c)
{
    std::string why;
    const Rccl *R = rccl(why);
    std::lock_guard<std::mutex> lock(c->mu);
    if (c->aborted) return;
    c->aborted = true;
    for (size_t i = 0; i < c->comms.size(); i++)
        if (R && R->CommAbort && c->comms[i]) { (void)R->CommAbort(c->comms[i]); c->comms[i] = nullptr; }
}

// rt_reduce_and_resolve; `entered` says whether this member's ncclReduce was issued
int reduce_and_resolve(rt_comm *c, int member, double *radiance_sum, int w, int h, int spp, int tone, uint8_t *bgr, void *stream, bool *entered)
{
    *entered = false;
    if (!c || member < 0 || member >= (int)c->comms.size()) return rt::set_error("rt_reduce_and_resolve: bad communicator member", -1);
    if (!radiance_sum || w <= 0 || h <= 0 || spp <= 0) return rt::set_error("rt_reduce_and_resolve: bad argument", -1);
    std::string why;
    const Rccl *R = rccl(why);
    if (!R) return rt::set_error(why, -3);
    ncclComm_t handle;
    {
        std::lock_guard<std::mutex> lock(c->mu);
        if (c->aborted) return rt::set_error("rt_reduce_and_resolve: the communicator was aborted after a member failed", -3);
        handle = c->comms[member];
    }
    hipError_t he = hipSetDevice(c->devices[member]);
    if (he != hipSuccess) return rt::set_error(std::string("hipSetDevice: ") + hipGetErrorString(he), -2);
    const bool root = c->rank0 + member == 0;
    // a direct reduce onto the root: one 8 w h 3 byte buffer per frame (25 MB at 1024 x 1024); on a
    // point-to-point fabric the root takes its peers' buffers over separate links
    *entered = true;
    const ncclResult_t e = R->Reduce(radiance_sum, radiance_sum, (size_t)w * (size_t)h * 3u, ncclFloat64, ncclSum, 0, handle, (hipStream_t)stream);
    if (e != ncclSuccess) return fail_nccl(R, "ncclReduce", e);
    if (root && bgr) return rt_resolve_tone_bgr8(radiance_sum, w, h, spp, tone, bgr, stream);
    return 0;
}

} // namespace

extern "C" {

int rt_comm_create(int ndev, const int *devices, rt_comm **out)
{
    if (ndev <= 0 || !devices || !out) return rt::set_error("rt_comm_create: bad argument", -1);
    for (int i = 0; i < ndev; i++) for (int j = 0; j < i; j++)
        if (devices[i] == devices[j]) return rt::set_error("rt_comm_create: a device is listed twice (RCCL takes one rank per device)", -1);
    std::string why;
    const Rccl *R = rccl(why);
    if (!R) return rt::set_error(why, -3);
    rt_comm *c = new rt_comm;
    c->nranks = ndev; c->rank0 = 0;
    c->devices.assign(devices, devices + ndev);
    c->comms.assign(ndev, nullptr);
    const ncclResult_t e = R->CommInitAll(c->comms.data(), ndev, devices);
    if (e != ncclSuccess) { delete c; return fail_nccl(R, "ncclCommInitAll", e); }
    *out = c;
    return 0;
}

int rt_comm_unique_id(uint8_t id[128])
{
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    if (!id) return rt::set_error("rt_comm_unique_id: null argument", -1);
    std::string why;
    const Rccl *R = rccl(why);
    if (!R) return rt::set_error(why, -3);
    ncclUniqueId u;
    const ncclResult_t e = R->GetUniqueId(&u);
    if (e != ncclSuccess) return fail_nccl(R, "ncclGetUniqueId", e);
    std::memcpy(id, &u, 128);
    return 0;
}

int rt_comm_init_rank(const uint8_t id[128], int nranks, int rank, int device, rt_comm **out)
{
    if (!id || !out || nranks <= 0 || rank < 0 || rank >= nranks) return rt::set_error("rt_comm_init_rank: bad argument", -1);
    std::string why;
    const Rccl *R = rccl(why);
    if (!R) return rt::set_error(why, -3);
    hipError_t he = hipSetDevice(device);
    if (he != hipSuccess) return rt::set_error(std::string("hipSetDevice: ") + hipGetErrorString(he), -2);
    ncclUniqueId u;
    std::memcpy(&u, id, 128);
    rt_comm *c = new rt_comm;
    c->nranks = nranks; c->rank0 = rank;
    c->devices.assign(1, device);
    c->comms.assign(1, nullptr);
    const ncclResult_t e = R->CommInitRank(&c->comms[0], nranks, u, rank);
    if (e != ncclSuccess) { delete c; return fail_nccl(R, "ncclCommInitRank", e); }
    *out = c;
    return 0;
}

int rt_comm_destroy(rt_comm *c)
{
    if (!c) return 0;
    std::string why;
    const Rccl *R = rccl(why);
    for (size_t i = 0; i < c->comms.size(); i++)
        if (R && c->comms[i]) { (void)hipSetDevice(c->devices[i]); (void)R->CommDestroy(c->comms[i]); }
    delete c;
    return 0;
}

int rt_comm_size(const rt_comm *c) { return c ? c->nranks : rt::set_error("rt_comm_size: null communicator", -1); }

int rt_reduce_and_resolve(rt_comm *c, int member, double *radiance_sum, int w, int h, int spp, int tone, uint8_t *bgr, void *stream)
{
    bool entered;
    return reduce_and_resolve(c, member, radiance_sum, w, h, spp, tone, bgr, stream, &entered);
}

} // extern "C"

// ---------------------------------------------------------------- several devices, one process
struct rt_multi {
    struct Member {
        int device = 0;
        rt_scene *scene = nullptr;
        double *sum = nullptr;           // device: w x h x 3
        uint64_t *counters = nullptr;    // device: 4
        uint8_t *bgr = nullptr;          // device, member 0 only
        size_t pixels = 0;
    };
    std::vector<Member> m;
    rt_comm *comm = nullptr;             // null: sums meet in host memory (reduce_mode 1)
};

namespace {

int multi_buffers(rt_multi::Member &q, size_t pixels, bool root)
{
    if (q.pixels == pixels && q.sum) return 0;
    if (q.sum) (void)hipFree(q.sum);
    if (q.bgr) (void)hipFree(q.bgr);
    q.sum = nullptr; q.bgr = nullptr; q.pixels = 0;
    hipError_t e = hipMalloc((void **)&q.sum, pixels * 3 * sizeof(double));
    if (e == hipSuccess && root) e = hipMalloc((void **)&q.bgr, pixels * 3);
    if (e == hipSuccess && !q.counters) e = hipMalloc((void **)&q.counters, 32);
    if (e != hipSuccess) return rt::set_error(std::string("hipMalloc (frame buffers): ") + hipGetErrorString(e), -2);
    q.pixels = pixels;
    return 0;
}

} // namespace

extern "C" {

int rt_multi_create(const rt_scene_desc *desc, const int *devices, int ndev, int reduce_mode, rt_multi **out)
{
    if (!desc || !devices || ndev <= 0 || !out) return rt::set_error("rt_multi_create: bad argument", -1);
    if (reduce_mode != RT_REDUCE_RCCL && reduce_mode != RT_REDUCE_HOST) return rt::set_error("rt_multi_create: unknown reduce mode", -1);
    rt_multi *M = new rt_multi;
    M->m.resize(ndev);
    // the octrees are built once; every device then takes its copy of the scene, side by side
    std::vector<rt_scene *> scenes(ndev, nullptr);
    int rc = rt::create_scenes(desc, devices, ndev, scenes.data());
    for (int i = 0; i < ndev; i++) { M->m[i].device = devices[i]; M->m[i].scene = scenes[i]; }
    if (rc == 0 && reduce_mode == RT_REDUCE_RCCL) rc = rt_comm_create(ndev, devices, &M->comm);
    if (rc) { const std::string keep = rt_last_error(); rt_multi_destroy(M); return rt::set_error(keep, rc); }
    *out = M;
    return 0;
}

int rt_multi_destroy(rt_multi *M)
{
    if (!M) return 0;
    for (auto &q : M->m) {
        (void)hipSetDevice(q.device);
        if (q.sum) (void)hipFree(q.sum);
        if (q.bgr) (void)hipFree(q.bgr);
        if (q.counters) (void)hipFree(q.counters);
        if (q.scene) rt_scene_destroy(q.scene);
    }
    if (M->comm) rt_comm_destroy(M->comm);
    delete M;
    return 0;
}

int rt_multi_size(const rt_multi *M) { return M ? (int)M->m.size() : rt::set_error("rt_multi_size: null handle", -1); }

rt_scene *rt_multi_scene(rt_multi *M, int member)
{
    if (!M || member < 0 || member >= (int)M->m.size()) { rt::set_error("rt_multi_scene: bad member", -1); return nullptr; }
    return M->m[member].scene;
}

int rt_multi_render(rt_multi *M, const rt_camera *cam, const rt_render_params *params, int tone,
                    double *radiance_sum_host, uint8_t *bgr_host, uint64_t counters[4])
{
    if (!M || !cam || !params) return rt::set_error("rt_multi_render: null argument", -1);
    if (params->width <= 0 || params->height <= 0 || params->n <= 0 || params->m <= 0) return rt::set_error("rt_multi_render: bad image or table size", -1);
    if (params->sample_begin < 0 || params->sample_begin > params->sample_end) return rt::set_error("rt_multi_render: bad sample range", -1);
    const int n = (int)M->m.size();
    const size_t pixels = (size_t)params->width * (size_t)params->height;
    const int total = params->sample_end - params->sample_begin, spp = params->n * params->m;
    std::vector<int> rcs(n, 0);
    std::vector<std::string> errs(n);
    // One host thread per device, in two rounds.  First the buffers: a member that cannot get its
    // buffers must be known before anybody enters the collective, which it could not join.
    auto prepare = [&](int i) {
        rt_multi::Member &q = M->m[i];
        auto done = [&](int rc) { rcs[i] = rc; if (rc) errs[i] = rt_last_error(); };
        if (int rc = rt_set_device(q.device)) return done(rc);
        if (int rc = multi_buffers(q, pixels, i == 0)) return done(rc);
        hipError_t e = hipMemsetAsync(q.sum, 0, pixels * 3 * sizeof(double), nullptr);
        if (e == hipSuccess) e = hipMemsetAsync(q.counters, 0, 32, nullptr);
        if (e != hipSuccess) return done(rt::set_error(std::string("hipMemsetAsync: ") + hipGetErrorString(e), -2));
        done(0);
    };
    {
        std::vector<std::thread> threads;
        for (int i = 1; i < n; i++) threads.emplace_back(prepare, i);
        prepare(0);
        for (auto &t : threads) t.join();
        for (int i = 0; i < n; i++) if (rcs[i]) return rt::set_error(errs[i], rcs[i]);
    }
    // Then this member's slice of the sample indices (contiguous, balanced) and -- with RCCL -- the
    // reduce, which every member enters.
    auto work = [&](int i) {
        rt_multi::Member &q = M->m[i];
        auto done = [&](int rc) { rcs[i] = rc; if (rc) errs[i] = rt_last_error(); };
        int rc0 = rt_set_device(q.device);
        hipError_t e = hipSuccess;
        rt_render_params p = *params;
        const int base = total / n, rem = total % n;
        p.sample_begin = params->sample_begin + i * base + (i < rem ? i : rem);
        p.sample_end = p.sample_begin + base + (i < rem ? 1 : 0);
        int rc = rc0 ? rc0 : rt_render(q.scene, cam, &p, q.sum, q.counters, nullptr);
        if (rc == 0) rc = rt_scene_check(q.scene, nullptr);
        // a member that failed still enters the reduce (its partial buffer is discarded with the frame):
        // leaving the others waiting in the collective would hang them; one that cannot even enter it
        // aborts the communicator, which releases them with an error
        if (M->comm) {
            bool entered = false;
            const std::string keep = rc ? rt_last_error() : "";
            const int rr = reduce_and_resolve(M->comm, i, q.sum, params->width, params->height, spp, tone, i == 0 ? q.bgr : nullptr, nullptr, &entered);
            if (!entered) abort_comm(M->comm);
            if (rc == 0) rc = rr; else rt::set_error(keep, rc);
        }
        e = hipDeviceSynchronize();
        if (rc == 0 && e != hipSuccess) rc = rt::set_error(std::string("hipDeviceSynchronize: ") + hipGetErrorString(e), -2);
        done(rc);
    };
    std::vector<std::thread> threads;
    for (int i = 1; i < n; i++) threads.emplace_back(work, i);
    work(0);
    for (auto &t : threads) t.join();
    for (int i = 0; i < n; i++) if (rcs[i]) return rt::set_error(errs[i], rcs[i]);
    // counters of all members; without RCCL the sums meet in host memory, in member order
    if (counters) {
        for (int i = 0; i < n; i++) {
            uint64_t c[4];
            (void)hipSetDevice(M->m[i].device);
            hipError_t e = hipMemcpy(c, M->m[i].counters, 32, hipMemcpyDeviceToHost);
            if (e != hipSuccess) return rt::set_error(std::string("hipMemcpy: ") + hipGetErrorString(e), -2);
            for (int k = 0; k < 4; k++) counters[k] += c[k];
        }
    }
    rt_multi::Member &root = M->m[0];
    if (int rc = rt_set_device(root.device)) return rc;
    if (!M->comm) {
        std::vector<double> acc(pixels * 3, 0.0), part(pixels * 3);
        for (int i = 0; i < n; i++) {
            (void)hipSetDevice(M->m[i].device);
            hipError_t e = hipMemcpy(part.data(), M->m[i].sum, pixels * 3 * sizeof(double), hipMemcpyDeviceToHost);
            if (e != hipSuccess) return rt::set_error(std::string("hipMemcpy: ") + hipGetErrorString(e), -2);
            for (size_t k = 0; k < acc.size(); k++) acc[k] += part[k];
        }
        (void)hipSetDevice(root.device);
        hipError_t e = hipMemcpy(root.sum, acc.data(), pixels * 3 * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) return rt::set_error(std::string("hipMemcpy: ") + hipGetErrorString(e), -2);
        if (int rc = rt_resolve_tone_bgr8(root.sum, params->width, params->height, spp, tone, root.bgr, nullptr)) return rc;
        e = hipDeviceSynchronize();
        if (e != hipSuccess) return rt::set_error(std::string("hipDeviceSynchronize: ") + hipGetErrorString(e), -2);
    }
    if (radiance_sum_host) {
        hipError_t e = hipMemcpy(radiance_sum_host, root.sum, pixels * 3 * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) return rt::set_error(std::string("hipMemcpy: ") + hipGetErrorString(e), -2);
    }
    if (bgr_host) {
        hipError_t e = hipMemcpy(bgr_host, root.bgr, pixels * 3, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return rt::set_error(std::string("hipMemcpy: ") + hipGetErrorString(e), -2);
    }
    return 0;
}

} // extern "C"
