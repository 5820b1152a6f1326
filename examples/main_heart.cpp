// An implicit surface scene written against include/rt_scene.hpp: the reference's HeartShape
// (Surface.h:151-174, intersected through its voxel octree) shown with the reference's
// NormalCheckMaterial (material.h:70-80), a SurfaceOfGenus2 with a Phong + mirror material and a
// Monte-Carlo diffuse heart, on a ground plane under a point Light and a FaceLight.
//   usage: main_heart [width height n min_size out.bmp]     (defaults 320 240 3 1e-2 heart.bmp)
#include <cstdio>
#include <cstdlib>
#include <memory>

#include "rt_scene.hpp"
using namespace rt_host;

int main(int argc, char **argv)
{
    const int px = argc > 1 ? std::atoi(argv[1]) : 320, py = argc > 2 ? std::atoi(argv[2]) : 240;
    const int n = argc > 3 ? std::atoi(argv[3]) : 3;
    const double min_size = argc > 4 ? std::atof(argv[4]) : 1e-2;
    const char *out = argc > 5 ? argv[5] : "heart.bmp";

    auto group = std::make_shared<Group>();
    auto add = [&](std::shared_ptr<Drawable> d, std::shared_ptr<Material> m) { d->setMaterial(std::move(m)); group->addChild(std::move(d)); };
    add(std::make_shared<HeartShape>(Vec3(-1.6, 1.6, 0.2), 1.1), std::make_shared<NormalCheckMaterial>());
    add(std::make_shared<HeartShape>(Vec3(1.9, 1.2, -0.6), 0.8), std::make_shared<Material>(Color(0, 0, 0, 0), Color(.75, .25, .25, 0), Material::NORMAL));
    add(std::make_shared<SurfaceOfGenus2>(Vec3(0.3, 1.1, 1.8), 0.9),
        std::make_shared<Material>(Color(.05, .08, .05, 0), Color(.3, .6, .35, 0), Color(.4, .4, .4, 0), Color(.3, .3, .3, 0), 20.0));
    auto grey = std::make_shared<Material>(Color(.06, .06, .06, 0), Color(.6, .6, .6, 0), Color(.1, .1, .1, 0), 4.0);
    add(std::make_shared<Ground>(0.0), grey);
    add(std::make_shared<Wall_z>(-3.5), grey);

    auto scene = std::make_shared<Scene>();
    scene->setPixels(px, py);
    scene->setSampleNum(n);
    scene->setImplicitMinSize(min_size);
    scene->setClearColor(Vec4(0.06, 0.07, 0.1, 1.0));
    scene->setRootNode(group);
    scene->setCamera(std::make_shared<Camera>(Vec3(0.6, 3.0, 8.5), Vec3(0.1, 1.3, 0), Vec3(0, 1, 0), 40, px * 1.0 / py));
    scene->addLight(std::make_shared<Light>(Vec3(3, 7, 6), Color(1, 1, 1, 0), Color(.9, .9, .8, 0), Color(1, 1, 1, 0)));
    scene->addLight(std::make_shared<FaceLight>(Vec3(0, 5.5, 2), Vec3(0, -1, 0), Vec3(0, 0, 1), 3.0, 3.0,
                                                Color(.1, .1, .1, 0), Color(.6, .6, .55, 0), Color(.3, .3, .3, 0)));
    if (scene->runParallelly(8)) { std::fprintf(stderr, "render failed: %s\n", scene->lastError()); return 1; }
    if (scene->saveToBpm(out)) { std::fprintf(stderr, "saveToBpm failed: %s\n", scene->lastError()); return 1; }
    std::printf("%dx%d @ %d spp, %llu rays -> %s\n", px, py, n * n, (unsigned long long)scene->raysTraced(), out);
    return 0;
}
