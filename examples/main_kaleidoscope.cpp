// The demo scene of the reference's driver (main.cpp:29-108: two huge mirror spheres and
// three coloured, mirror-like SphereLights forming a kaleidoscope), written against
// include/rt_scene.hpp and rendered on the GPU.
//   usage: main_kaleidoscope [width height n out.bmp]     (defaults 1020 630 20 test.bmp)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>

#include "rt_scene.hpp"
using namespace rt_host;

int main(int argc, char **argv)
{
    const int px = argc > 1 ? std::atoi(argv[1]) : 1020, py = argc > 2 ? std::atoi(argv[2]) : 630;
    const int n = argc > 3 ? std::atoi(argv[3]) : 20;
    const char *out = argc > 4 ? argv[4] : "test.bmp";
    const double PI = 3.1415926;
    double R = 300, T = 30 * PI / 180., D = R / std::cos(T), Z = 62;
    Color r(1.2, 1, 1, 1), g(1, 1.1, 1, 1), b(1, 1, 1, 1);

    auto group = std::make_shared<Group>();
    auto back = std::make_shared<Sphere>(Vec3(50, 28, Z) + Vec3(0, 0, -1) * (R * 2 * std::sqrt(2. / 3.)), R);
    auto front = std::make_shared<Sphere>(Vec3(50, 28, Z) + Vec3(0, 0, -R * 2 * std::sqrt(2. / 3.) / 3.),
                                          2 * 2 * R * 2 * std::sqrt(2. / 3.) - R * 2 * std::sqrt(2. / 3.) / 3.);
    group->addChild(back);
    group->addChild(front);
    Color C(0.3, 0.7, 0.8, 1);
    double emission = 0.8e-1;
    auto mr = std::make_shared<Material>(C * emission & r, Color(1.15, .996, .996, 0), Material::SPECULAR);
    auto mg = std::make_shared<Material>(C * emission & g, Color(.99, 1.03, .99, 0), Material::SPECULAR);
    auto mb = std::make_shared<Material>(C * emission & b, Color(.996, .996, .996, 0), Material::SPECULAR);
    auto mback = std::make_shared<Material>(C * 0e-2, Color(.95, .95, .95, 0), Material::SPECULAR);
    auto mfront = std::make_shared<Material>(C * 0e-2, Color(.93, .93, .93, 0), Material::SPECULAR);

    auto scene = std::make_shared<Scene>();
    scene->setPixels(px, py);
    scene->setSampleNum(n);
    scene->setClearColor(Vec4(0.0, 0.0, 0.0, 1.0));
    auto camera = std::make_shared<Camera>(Vec3(50, 40, -120), Vec3(50, 40, Z), Vec3(0, 1, 0), 95, px * 1.0 / py);
    auto lr = std::make_shared<SphereLight>(Vec3(50, 28, Z) + Vec3(std::cos(T), std::sin(T), 0) * D, R, C * emission & r);
    auto lg = std::make_shared<SphereLight>(Vec3(50, 28, Z) + Vec3(-std::cos(T), std::sin(T), 0) * D, R, C * emission & g);
    auto lb = std::make_shared<SphereLight>(Vec3(50, 28, Z) + Vec3(0, -1, 0) * D, R, C * emission & b);
    back->setMaterial(mback);
    front->setMaterial(mfront);
    lr->getGeometry()->setMaterial(mr);
    lg->getGeometry()->setMaterial(mg);
    lb->getGeometry()->setMaterial(mb);
    scene->setRootNode(group);
    scene->setCamera(camera);
    scene->addLight(lr);
    scene->addLight(lg);
    scene->addLight(lb);

    if (scene->runParallelly(8)) { std::fprintf(stderr, "render failed: %s\n", scene->lastError()); return 1; }
    if (scene->saveToBpm(out)) { std::fprintf(stderr, "saveToBpm failed: %s\n", scene->lastError()); return 1; }
    Color c;
    scene->queryColor(px / 2, py / 2, c);
    std::printf("%dx%d @ %d spp, %llu rays, centre pixel %.6f %.6f %.6f -> %s\n", px, py, n * n,
                (unsigned long long)scene->raysTraced(), c.x_, c.y_, c.z_, out);
    return 0;
}
