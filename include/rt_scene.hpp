// rt_scene.hpp -- host-side scene API with the reference's class and method names, on top of
// the C ABI in rt_hip.h.  A driver written against the reference (main.cpp:29-108) ports by
// changing its includes: build Group / Sphere / Material / Camera / SphereLight objects, hand
// them to a Scene, call runParallelly(), read pixels with queryColor() or saveToBpm().
//
// What differs from the reference, on purpose:
//   * Scene::runParallelly(threadNum) renders on GPUs: the reference's worker count becomes the
//     number of devices to use at most (one host thread, one copy of the scene and one slice of
//     the sample indices per device, the devices' sums added with one RCCL reduce); 0 or 1, or a
//     machine with one GPU, means one device.  There is no CPU rendering path behind this header.
//   * Mesh is new: a Drawable holding triangles in an Octree (the role HeartShape /
//     ImplicitSurface play in the reference, Surface.h:131-174).
//   * the sampler constants that are private members without setters in the reference
//     (scene.h:52-54) are settable: setSampleNum, setMaxJump, setRusJump, setSeed, setFaceForward.
//   * both shading branches of Scene::rayColor are on the device: materials with MONTECARLO
//     (scene.cpp:205-304) and the Phong / Whitted ones without it (scene.cpp:305-375).
// Header-only; link with librt_hip.so.
#pragma once
#include <cmath>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "rt_hip.h"

namespace rt_host {

// types.h:8-93, 97-178
struct Vec3 {
    double x_ = 0, y_ = 0, z_ = 0;
    Vec3() = default;
    Vec3(double x, double y, double z) : x_(x), y_(y), z_(z) {}
    Vec3 operator+(const Vec3 &r) const { return Vec3(x_ + r.x_, y_ + r.y_, z_ + r.z_); }
    Vec3 operator-(const Vec3 &r) const { return Vec3(x_ - r.x_, y_ - r.y_, z_ - r.z_); }
    Vec3 operator*(double s) const { return Vec3(x_ * s, y_ * s, z_ * s); }
    Vec3 operator^(const Vec3 &r) const { return Vec3(y_ * r.z_ - r.y_ * z_, z_ * r.x_ - x_ * r.z_, x_ * r.y_ - r.x_ * y_); }
    double length() const { return std::sqrt(x_ * x_ + y_ * y_ + z_ * z_); }
    Vec3 normalize() const { return *this * (1 / length()); }
};
struct Vec4 {
    double x_ = 0, y_ = 0, z_ = 0, w_ = 0;
    Vec4() = default;
    Vec4(double x, double y, double z, double w) : x_(x), y_(y), z_(z), w_(w) {}
    Vec4 operator*(double s) const { return Vec4(x_ * s, y_ * s, z_ * s, w_ * s); }
    Vec4 operator&(const Vec4 &r) const { return Vec4(x_ * r.x_, y_ * r.y_, z_ * r.z_, w_ * r.w_); }
};
using Color = Vec4;

// material.h:7-67
class Material {
public:
    enum Type { NULLTYPE = 0x0, NORMAL = 0x1, SPECULAR = 0x2, TRANSPARENT = 0x4, MONTECARLO = 0x8 };
    Material(Color amb, Color dif, Color spe, double shi)                    // material.cpp:2-6
    {
        rec_ = rt_material{};
        rec_.type = NORMAL;
        set3(rec_.ambient, amb); set3(rec_.color, dif); set3(rec_.specular, spe); rec_.shine = shi;
    }
    Material(Color amb, Color dif, Color spe, Color mir, double shi)         // material.cpp:8-12
    {
        rec_ = rt_material{};
        rec_.type = SPECULAR | NORMAL;
        set3(rec_.ambient, amb); set3(rec_.color, dif); set3(rec_.specular, spe); set3(rec_.mirror, mir); rec_.shine = shi;
    }
    Material(Color emission, Color color, Type type)                         // material.cpp:14-17
    {
        rec_ = rt_material{};
        rec_.type = (int)type | MONTECARLO;
        set3(rec_.emission, emission); set3(rec_.color, color);
    }
    Material(Color a, double nr)                                             // material.cpp:19-23
    {
        rec_ = rt_material{};
        rec_.type = TRANSPARENT;
        set3(rec_.atten, a); rec_.nr = nr;
    }
    void setMonteCarlo(Color color = Color(-1, -1, -1, 0), Color emission = Color(-1, -1, -1, 0))   // material.cpp:40-47
    {
        rec_.type |= MONTECARLO;
        if (color.x_ >= 0) set3(rec_.color, color);
        if (emission.x_ >= 0) set3(rec_.emission, emission);
    }
    void setMirrorBlur(double b) { rec_.blur = b; }
protected:
    void setCheck(int kind) { rec_.check = kind; }
    void setPerlin(int kind, double scale) { rec_.perlin = kind; rec_.perlin_scale = scale; }
    void setCheckBox(const Vec3 &lo, const Vec3 &hi)
    {
        rec_.check_box[0] = lo.x_; rec_.check_box[1] = lo.y_; rec_.check_box[2] = lo.z_;
        rec_.check_box[3] = hi.x_; rec_.check_box[4] = hi.y_; rec_.check_box[5] = hi.z_;
    }
public:
    int getType() const { return rec_.type; }
    const rt_material &record() const { return rec_; }
private:
    static void set3(double *d, const Color &c) { d[0] = c.x_; d[1] = c.y_; d[2] = c.z_; }
    rt_material rec_;
};

// material.h:70-92: debug materials whose ambient colour shows the normal / the local position
class NormalCheckMaterial : public Material {
public:
    NormalCheckMaterial() : Material(Color(0, 0, 0, 1), Color(0, 0, 0, 1), Color(0, 0, 0, 1), 1) { setCheck(RT_CHECK_NORMAL); }
};
class PositionCheckMaterial : public Material {
public:
    PositionCheckMaterial(Vec3 lo, Vec3 hi) : Material(Color(0, 0, 0, 1), Color(0, 0, 0, 1), Color(0, 0, 0, 1), 1)
    {
        setCheck(RT_CHECK_POSITION); setCheckBox(lo, hi);
    }
};

// material.h:95-164: Phong materials modulated by Perlin noise at the hit position
class PerlinNoiseNormalMaterial : public Material {
public:
    PerlinNoiseNormalMaterial(Color amb, Color dif, Color spe, double shi, double scale = 1) : Material(amb, dif, spe, shi) { setPerlin(RT_PERLIN_NORMAL, scale); }
protected:
    PerlinNoiseNormalMaterial(Color amb, Color dif, Color spe, double shi, double scale, int kind) : Material(amb, dif, spe, shi) { setPerlin(kind, scale); }
};
class PerlinNoiseNormalMaterial_Blood : public PerlinNoiseNormalMaterial {
public:
    PerlinNoiseNormalMaterial_Blood(Color amb, Color dif, Color spe, double shi, double scale = 1)
        : PerlinNoiseNormalMaterial(amb, dif, spe, shi, scale, RT_PERLIN_BLOOD) {}
};

// Surface.h:47-74
class Node {
public:
    virtual ~Node() = default;
    int setMaterial(std::shared_ptr<Material> m) { mat_ = std::move(m); return 0; }
    std::shared_ptr<Material> getMaterial() const { return mat_; }
    // appends this node (and its children) to a flat description; returns -1 on error
    virtual int flatten(std::vector<rt_object> &objs, std::vector<rt_material> &mats,
                        std::vector<std::vector<double>> &meshes) const = 0;
protected:
    int material_index(std::vector<rt_material> &mats) const
    {
        if (!mat_) return -1;
        mats.push_back(mat_->record());
        return (int)mats.size() - 1;
    }
    std::shared_ptr<Material> mat_;
};
class Drawable : public Node {};

class Sphere : public Drawable {                                             // Surface.h:95-111
public:
    Sphere(Vec3 center, double radius) : c_(center), r_(radius) {}
    int flatten(std::vector<rt_object> &objs, std::vector<rt_material> &mats, std::vector<std::vector<double>> &) const override
    {
        rt_object o{}; o.type = RT_OBJ_SPHERE; o.mesh = -1;
        if ((o.material = material_index(mats)) < 0) return -1;
        o.p[0] = c_.x_; o.p[1] = c_.y_; o.p[2] = c_.z_; o.p[3] = r_;
        objs.push_back(o);
        return 0;
    }
private:
    Vec3 c_; double r_;
};
template <int KIND> class Plane_ : public Drawable {                        // Surface.h:176-214
public:
    explicit Plane_(double coord) : c_(coord) {}
    int flatten(std::vector<rt_object> &objs, std::vector<rt_material> &mats, std::vector<std::vector<double>> &) const override
    {
        rt_object o{}; o.type = KIND; o.mesh = -1;
        if ((o.material = material_index(mats)) < 0) return -1;
        o.p[0] = c_;
        objs.push_back(o);
        return 0;
    }
private:
    double c_;
};
// HeartShape(center, scale), Surface.h:151-174; SurfaceOfGenus2(center, scale), Surface.h:129-147:
// implicit surfaces intersected through a voxel octree that the library builds at upload
template <int KIND> class Implicit_ : public Drawable {
public:
    Implicit_(Vec3 center, double scale) : c_(center), s_(scale) {}
    int flatten(std::vector<rt_object> &objs, std::vector<rt_material> &mats, std::vector<std::vector<double>> &) const override
    {
        rt_object o{}; o.type = KIND; o.mesh = -1;
        if ((o.material = material_index(mats)) < 0) return -1;
        o.p[0] = c_.x_; o.p[1] = c_.y_; o.p[2] = c_.z_; o.p[3] = s_;
        objs.push_back(o);
        return 0;
    }
private:
    Vec3 c_; double s_;
};
using HeartShape = Implicit_<RT_OBJ_HEART>;
using SurfaceOfGenus2 = Implicit_<RT_OBJ_GENUS2>;
using Ground = Plane_<RT_OBJ_GROUND>;
using Wall_z = Plane_<RT_OBJ_WALL_Z>;
using Wall_x = Plane_<RT_OBJ_WALL_X>;

// New: triangles (v0 v1 v2, 9 doubles each) held in an octree built with the reference policy.
class Mesh : public Drawable {
public:
    explicit Mesh(std::vector<double> tris) : tris_(std::move(tris)) {}
    void addTriangle(Vec3 a, Vec3 b, Vec3 c)
    {
        for (const Vec3 *v : {&a, &b, &c}) { tris_.push_back(v->x_); tris_.push_back(v->y_); tris_.push_back(v->z_); }
    }
    int flatten(std::vector<rt_object> &objs, std::vector<rt_material> &mats, std::vector<std::vector<double>> &meshes) const override
    {
        if (tris_.size() % 9) return -1;
        rt_object o{}; o.type = RT_OBJ_MESH;
        if ((o.material = material_index(mats)) < 0) return -1;
        meshes.push_back(tris_);
        o.mesh = (int)meshes.size() - 1;
        objs.push_back(o);
        return 0;
    }
private:
    std::vector<double> tris_;
};

class Group : public Node {                                                  // Surface.h:76-86
public:
    void addChild(std::shared_ptr<Node> child) { children_.push_back(std::move(child)); }
    int flatten(std::vector<rt_object> &objs, std::vector<rt_material> &mats, std::vector<std::vector<double>> &meshes) const override
    {
        for (const auto &c : children_) if (c->flatten(objs, mats, meshes)) return -1;
        return 0;
    }
private:
    std::vector<std::shared_ptr<Node>> children_;
};

class Camera {                                                               // camera.h:4-28
public:
    Camera(Vec3 lookFrom, Vec3 lookAt, Vec3 vup, double vfov, double aspect)
    {
        const double f[3] = {lookFrom.x_, lookFrom.y_, lookFrom.z_}, a[3] = {lookAt.x_, lookAt.y_, lookAt.z_}, u[3] = {vup.x_, vup.y_, vup.z_};
        rt_camera_init(f, a, u, vfov, aspect, &rec_);
    }
    virtual ~Camera() = default;
    virtual void setTime(double) {}                                          // RTObject.h:110-117
    const rt_camera &record() const { return rec_; }
protected:
    rt_camera rec_;
};
class ChangeFovCamera : public Camera {                                      // camera.cpp:74-94
public:
    ChangeFovCamera(Vec3 lookFrom, Vec3 lookAt, Vec3 vup, double vfov, double aspect)
        : Camera(lookFrom, lookAt, vup, vfov, aspect), origin_(rec_) {}
    void setTime(double time) override { rt_camera_change_fov(&origin_, time, &rec_); }
private:
    rt_camera origin_;
};
class DepthCamera : public Camera {                                          // camera.cpp:50-72
public:
    DepthCamera(Vec3 lookFrom, Vec3 lookAt, Vec3 vup, double vfov, double aspect, double aperture)
        : Camera(lookFrom, lookAt, vup, vfov, aspect) { rec_.aperture = aperture; }
};

class Light {                                                                // light.h:8-30
public:
    Light() = default;
    Light(Vec3 pos, Color ambient, Color diffuse, Color specular)             // light.cpp:8-11
    {
        rec_.type = RT_LIGHT_POINT;
        rec_.pos[0] = pos.x_; rec_.pos[1] = pos.y_; rec_.pos[2] = pos.z_;
        set_colors(ambient, diffuse, specular);
    }
    virtual ~Light() = default;
    virtual std::shared_ptr<Drawable> getGeometry() { return nullptr; }
    virtual bool setTime(double) { return false; }                           // true if the record changed
    const rt_light &record() const { return rec_; }
protected:
    void set_colors(const Color &a, const Color &d, const Color &sp)
    {
        rec_.ambient[0] = a.x_; rec_.ambient[1] = a.y_; rec_.ambient[2] = a.z_;
        rec_.diffuse[0] = d.x_; rec_.diffuse[1] = d.y_; rec_.diffuse[2] = d.z_;
        rec_.specular[0] = sp.x_; rec_.specular[1] = sp.y_; rec_.specular[2] = sp.z_;
    }
    rt_light rec_{};
};
class SphereLight : public Light {                                           // light.cpp:109-116
public:
    SphereLight(Vec3 pos, double radius, Color emission)
    {
        rec_.type = RT_LIGHT_SPHERE;
        rec_.pos[0] = pos.x_; rec_.pos[1] = pos.y_; rec_.pos[2] = pos.z_; rec_.radius = radius;
        rec_.diffuse[0] = emission.x_; rec_.diffuse[1] = emission.y_; rec_.diffuse[2] = emission.z_;
        sphere_ = std::make_shared<Sphere>(pos, radius);
        sphere_->setMaterial(std::make_shared<Material>(emission, Color(), Material::NORMAL));
    }
    std::shared_ptr<Drawable> getGeometry() override { return sphere_; }
private:
    std::shared_ptr<Sphere> sphere_;
};
class FaceLight : public Light {                                             // light.cpp:50-56
public:
    FaceLight(Vec3 pos, Vec3 normal, Vec3 up, double width, double height, Color ambient, Color diffuse, Color specular)
    {
        rec_.type = RT_LIGHT_FACE;
        const Vec3 right = up ^ normal, n = normal.normalize();
        rec_.pos[0] = pos.x_; rec_.pos[1] = pos.y_; rec_.pos[2] = pos.z_;
        rec_.normal[0] = n.x_; rec_.normal[1] = n.y_; rec_.normal[2] = n.z_;
        rec_.up[0] = up.x_; rec_.up[1] = up.y_; rec_.up[2] = up.z_;
        rec_.right[0] = right.x_; rec_.right[1] = right.y_; rec_.right[2] = right.z_;
        rec_.width = width; rec_.height = height;
        set_colors(ambient, diffuse, specular);
    }
};

class MovableFaceLight : public FaceLight {                                 // light.cpp:87-107
public:
    MovableFaceLight(Vec3 pos, Vec3 normal, Vec3 up, double width, double height, Color ambient, Color diffuse, Color specular)
        : FaceLight(pos, normal, up, width, height, ambient, diffuse, specular), origin_(rec_) {}
    bool setTime(double time) override { return rt_light_move_face(&origin_, time, &rec_) == 0; }
private:
    rt_light origin_;
};

class Scene {                                                                // scene.h:8-56
public:
    Scene() = default;
    ~Scene() { release(); }
    Scene(const Scene &) = delete;
    Scene &operator=(const Scene &) = delete;

    int setPixels(int x, int y)                                              // scene.cpp:52-58
    {
        if (x <= 0 || y <= 0) return -1;
        px_ = x; py_ = y; sum_.clear(); run_ = false;
        return 0;
    }
    int getPixels(int *x, int *y) const { if (px_ == -1 || py_ == -1) return -1; *x = px_; *y = py_; return 0; }
    void setRootNode(std::shared_ptr<Node> node) { node_ = std::move(node); dirty_ = true; }   // scene.cpp:68-73
    void setCamera(std::shared_ptr<Camera> camera) { camera_ = std::move(camera); }
    void setClearColor(Color c) { background_ = c; dirty_ = true; }
    void addLight(std::shared_ptr<Light> light) { lights_.push_back(std::move(light)); dirty_ = true; }   // scene.cpp:86-93
    // made settable (private constants in the reference, scene.h:52-54; srand value scene.cpp:22)
    void setSampleNum(int n, int m = -1) { n_ = n; m_ = m < 0 ? n : m; }
    void setMaxJump(int j) { maxJump_ = j; }
    void setRusJump(int j) { rusJump_ = j; }
    void setSeed(uint64_t s) { seed_ = s; }
    void setFaceForward(bool on) { faceForward_ = on; }
    void setOctree(int maxSize, int maxDepth) { octMax_ = maxSize; octDepth_ = maxDepth; dirty_ = true; }
    // tonereproductionmethod.h:11-35 (RT_TONE_SIGMOID / RT_TONE_SQRT), applied by saveToBpm and queryColor
    void setToneReproduction(int tone) { tone_ = tone; }
    void setImplicitMinSize(double s) { implicitMin_ = s; dirty_ = true; }      // octree.h:66 (3e-3 when 0)
    // RTObject::setTime cascade (RTObject.h:110-117, main.cpp:85): the camera and the lights follow
    // the clock; a moved light is patched into the uploaded scene, nothing is rebuilt.
    int setTime(double time)
    {
        // Perlin materials own generators whose vectors drift with time (algorithms.cpp:57-77)
        if (usesPerlin()) {
            ensurePerlin();
            rt_perlin_time_changed(&perlinClock_, time_, time, 256, perlinG_.data());
            if (!dirty_) for (rt_scene *h : handles()) if (rt_scene_set_perlin(h, perlinP_.data(), perlinG_.data(), 256, nullptr)) return -1;
        }
        time_ = time;
        if (camera_) camera_->setTime(time);
        for (size_t i = 0; i < lights_.size(); i++)
            if (lights_[i]->setTime(time) && !dirty_)
                for (rt_scene *h : handles()) if (rt_scene_set_light(h, i, &lights_[i]->record(), nullptr)) return -1;
        return 0;
    }

    // Scene::run() / runParallelly(threadNum), scene.cpp:95-172: the whole frame, on the GPU(s).
    int run() { return runParallelly(1); }
    int runParallelly(int threadNum)
    {
        if (px_ <= 0 || py_ <= 0 || !camera_) return -1;
        int ndev = rt_device_count();
        if (ndev < 1) return -1;
        if (threadNum < ndev) ndev = threadNum < 1 ? 1 : threadNum;
        if (upload(ndev)) return -1;
        rt_render_params p{};
        p.width = px_; p.height = py_; p.n = n_; p.m = m_; p.sample_begin = 0; p.sample_end = n_ * m_;
        p.max_jump = maxJump_; p.rus_jump = rusJump_; p.seed = seed_; p.face_forward = faceForward_ ? 1 : 0;
        sum_.assign((size_t)px_ * py_ * 3, 0.0);
        if (multi_) {
            if (rt_multi_render(multi_, &camera_->record(), &p, tone_, sum_.data(), nullptr, rays_)) return -1;
        } else if (rt_render_host(handle_, &camera_->record(), &p, sum_.data(), rays_)) return -1;
        run_ = true;
        return 0;
    }
    // scene.h:72-80; the stored colour is the average, regularized (scene.cpp:112-113)
    int queryColor(unsigned x, unsigned y, Color &color) const
    {
        if (!run_ || x >= (unsigned)px_ || y >= (unsigned)py_) return -1;
        const double inv = 1.0 / (n_ * m_);
        const double *s = &sum_[((size_t)y * px_ + x) * 3];
        auto reg = [&](double v) {
            if (tone_ == RT_TONE_SIGMOID) v = 1.0 / (1 + std::exp(-(2 * v - 1))) * 2.164 - 0.582;
            else if (tone_ == RT_TONE_SQRT) v = v > 1 ? std::pow(v, 1.0 / 2) : v;
            return v > 1.0 ? 1.0 : (v < 0 ? 0.0 : v);
        };
        color = Color(reg(s[0] * inv), reg(s[1] * inv), reg(s[2] * inv), 0);
        return 0;
    }
    int saveToBpm(const char *filename) const                                // scene.cpp:174-192
    {
        if (!run_) return -1;
        std::vector<uint8_t> bgr((size_t)px_ * py_ * 3);
        if (rt_resolve_tone_bgr8_host(sum_.data(), px_, py_, n_ * m_, tone_, bgr.data())) return -1;
        return rt_save_bmp(bgr.data(), px_, py_, filename) ? -1 : 0;
    }
    const char *lastError() const { return rt_last_error(); }
    uint64_t raysTraced() const { return rays_[0]; }

private:
    void release()
    {
        if (handle_) { rt_scene_destroy(handle_); handle_ = nullptr; }
        if (multi_) { rt_multi_destroy(multi_); multi_ = nullptr; }
    }
    // the uploaded copies of the scene: one, or one per device
    std::vector<rt_scene *> handles() const
    {
        std::vector<rt_scene *> v;
        if (handle_) v.push_back(handle_);
        if (multi_) for (int i = 0; i < rt_multi_size(multi_); i++) v.push_back(rt_multi_scene(multi_, i));
        return v;
    }
    int upload(int ndev)
    {
        if (!dirty_ && ((ndev == 1 && handle_) || (ndev > 1 && multi_ && rt_multi_size(multi_) == ndev))) return 0;
        release();
        std::vector<rt_object> objs; std::vector<rt_material> mats; std::vector<std::vector<double>> meshes; std::vector<rt_light> lights;
        if (node_ && node_->flatten(objs, mats, meshes)) return -1;
        for (const auto &l : lights_) {                                      // light geometry joins the group, scene.cpp:89-91
            lights.push_back(l->record());
            if (auto g = l->getGeometry()) if (g->flatten(objs, mats, meshes)) return -1;
        }
        std::vector<rt_mesh> mrec(meshes.size());
        for (size_t i = 0; i < meshes.size(); i++) { mrec[i].tris = meshes[i].data(); mrec[i].ntris = meshes[i].size() / 9; }
        rt_scene_desc d{};
        d.objects = objs.data(); d.nobjects = objs.size();
        d.materials = mats.data(); d.nmaterials = mats.size();
        d.lights = lights.data(); d.nlights = lights.size();
        d.meshes = mrec.data(); d.nmeshes = mrec.size();
        d.background[0] = background_.x_; d.background[1] = background_.y_; d.background[2] = background_.z_;
        d.tri_box_pad = 1e-7; d.octree_max_size = octMax_; d.octree_overlap_tol = 3; d.octree_max_depth = octDepth_;
        d.implicit_min_size = implicitMin_;
        bool perlin = false;
        for (const rt_material &m : mats) perlin = perlin || m.perlin != RT_PERLIN_NONE;
        if (perlin) { ensurePerlin(); d.perlin_p = perlinP_.data(); d.perlin_g = perlinG_.data(); d.perlin_n = 256; }
        if (ndev > 1) {
            std::vector<int> devices(ndev);
            for (int i = 0; i < ndev; i++) devices[i] = i;
            // the members' sums meet through RCCL; where that cannot be set up (no loadable RCCL, a
            // communicator that will not form) they meet in host memory instead -- same frame, slower exchange
            if (rt_multi_create(&d, devices.data(), ndev, RT_REDUCE_RCCL, &multi_) &&
                rt_multi_create(&d, devices.data(), ndev, RT_REDUCE_HOST, &multi_)) return -1;
        } else if (rt_scene_create(&d, &handle_)) return -1;
        dirty_ = false;
        return 0;
    }
    std::shared_ptr<Node> node_;
    std::shared_ptr<Camera> camera_;
    std::vector<std::shared_ptr<Light>> lights_;
    Color background_;
    int px_ = -1, py_ = -1;
    bool run_ = false, dirty_ = true, faceForward_ = false;
    int n_ = 20, m_ = 20, maxJump_ = 40, rusJump_ = 12;                      // scene.h:52-54
    int octMax_ = 128, octDepth_ = 12;                                       // octree.h:50
    bool usesPerlin() const
    {
        std::vector<rt_object> objs; std::vector<rt_material> mats; std::vector<std::vector<double>> meshes;
        if (!node_ || node_->flatten(objs, mats, meshes)) return false;
        for (const rt_material &m : mats) if (m.perlin != RT_PERLIN_NONE) return true;
        return false;
    }
    void ensurePerlin()
    {
        if (!perlinP_.empty()) return;
        perlinP_.resize(256); perlinG_.resize(3 * 256);
        rt_perlin_tables(12325, 256, perlinP_.data(), perlinG_.data());      // SEED, N: algorithms.h:91,99
    }
    std::vector<int32_t> perlinP_;
    std::vector<double> perlinG_;
    rt_perlin_clock perlinClock_{0, 0, 0.0};
    double time_ = 0;
    double implicitMin_ = 0;
    int tone_ = RT_TONE_NONE;
    uint64_t seed_ = 1243;                                                   // scene.cpp:22
    rt_scene *handle_ = nullptr;       // one device
    rt_multi *multi_ = nullptr;        // several devices
    std::vector<double> sum_;
    uint64_t rays_[4] = {0, 0, 0, 0};
};

// AviSaver.h:6-24.  Frames go into a Motion-JPEG AVI through the library's own encoder (the
// reference hands them to OpenCV's); with loop > 1 the frames are kept and written `loop` times
// by Save().
class AviSaver {
public:
    AviSaver() = default;
    ~AviSaver() { if (avi_) rt_avi_close(avi_); }
    AviSaver(const AviSaver &) = delete;
    AviSaver &operator=(const AviSaver &) = delete;
    void Open(const std::string &path, int width, int height, double fps, unsigned int loop = 1)
    {
        if (avi_) { rt_avi_close(avi_); avi_ = nullptr; }
        rt_avi_open_mjpg(path.c_str(), width, height, fps, 95, &avi_);      // fourcc MJPG as AviSaver.cpp:60; 95 = OpenCV's default JPEG quality
        width_ = width; height_ = height; loop_ = (int)loop;
    }
    bool isOpen() const { return avi_ != nullptr; }
    // pdata: height rows of width B,G,R bytes, bottom row first (what main.cpp:87-99 fills in)
    void AddImg(const uint8_t *pdata)
    {
        if (!avi_) return;
        if (loop_ == 1) rt_avi_add_frame(avi_, pdata);
        else kept_.emplace_back(pdata, pdata + (size_t)width_ * height_ * 3);
    }
    void AddImg(const std::vector<uint8_t> &rgbs) { if (rgbs.size() >= (size_t)width_ * height_ * 3) AddImg(rgbs.data()); }
    void Save()                                                              // AviSaver.cpp:44-53; also finishes the file
    {
        if (!avi_) return;
        if (loop_ > 1) for (int i = 0; i < loop_; i++) for (const auto &f : kept_) rt_avi_add_frame(avi_, f.data());
        rt_avi_close(avi_);
        avi_ = nullptr;
    }
private:
    rt_avi *avi_ = nullptr;
    int width_ = 0, height_ = 0, loop_ = 1;
    std::vector<std::vector<uint8_t>> kept_;
};

} // namespace rt_host
