// The animation mode of the reference's driver (main.cpp:74-101: a loop over `time`, one frame
// per step, scene->setTime(time) before each render) written against include/rt_scene.hpp with
// the reference's animated classes: a ChangeFovCamera (camera.cpp:74-94) and a MovableFaceLight
// (light.cpp:87-107) over Phong / mirror / glass spheres.  Frames are written as BMP files
// <prefix>NNN.bmp and, as in the reference, into <prefix>anim.avi through AviSaver (Motion-JPEG
// by the library's own encoder; the reference's goes through OpenCV's).
//   usage: main_animation [width height n frames prefix]   (defaults 160 120 2 8 frame_)
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "rt_scene.hpp"
using namespace rt_host;

int main(int argc, char **argv)
{
    const int px = argc > 1 ? std::atoi(argv[1]) : 160, py = argc > 2 ? std::atoi(argv[2]) : 120;
    const int n = argc > 3 ? std::atoi(argv[3]) : 2, frames = argc > 4 ? std::atoi(argv[4]) : 8;
    const std::string prefix = argc > 5 ? argv[5] : "frame_";

    auto red = std::make_shared<Material>(Color(.10, .04, .04, 0), Color(.70, .25, .25, 0), Color(.5, .5, .5, 0), 24.0);
    auto chrome = std::make_shared<Material>(Color(.02, .02, .02, 0), Color(.10, .10, .12, 0), Color(.6, .6, .6, 0), Color(.75, .75, .8, 0), 60.0);
    auto glass = std::make_shared<Material>(Color(.02, .03, .05, 0), 1.5);
    auto floorm = std::make_shared<Material>(Color(.05, .05, .06, 0), Color(.50, .50, .60, 0), Color(.1, .1, .1, 0), 4.0);
    auto wallm = std::make_shared<Material>(Color(.05, .06, .05, 0), Color(.45, .55, .40, 0), Color(.2, .2, .2, 0), 8.0);

    auto group = std::make_shared<Group>();
    auto add = [&](std::shared_ptr<Drawable> d, std::shared_ptr<Material> m) { d->setMaterial(std::move(m)); group->addChild(std::move(d)); };
    add(std::make_shared<Sphere>(Vec3(0, 1, 0), 1.0), red);
    add(std::make_shared<Sphere>(Vec3(2.3, 1, 0.4), 1.0), chrome);
    add(std::make_shared<Sphere>(Vec3(-2.3, 1, 0.8), 1.0), glass);
    add(std::make_shared<Ground>(0.0), floorm);
    add(std::make_shared<Wall_z>(-4.0), wallm);
    add(std::make_shared<Wall_x>(-6.0), wallm);

    auto scene = std::make_shared<Scene>();
    scene->setPixels(px, py);
    scene->setSampleNum(n);
    scene->setClearColor(Vec4(0.05, 0.07, 0.1, 1.0));
    scene->setRootNode(group);
    scene->setCamera(std::make_shared<ChangeFovCamera>(Vec3(0.5, 2.8, 9), Vec3(0, 1.2, 0), Vec3(0, 1, 0), 30, px * 1.0 / py));
    scene->addLight(std::make_shared<Light>(Vec3(4, 7, 5), Color(.5, .5, .5, 0), Color(.9, .9, .8, 0), Color(1, 1, 1, 0)));
    scene->addLight(std::make_shared<MovableFaceLight>(Vec3(-3, 6, 3), Vec3(0.4, -1, -0.3), Vec3(0, 0, 1), 2.5, 1.5,
                                                       Color(.2, .2, .2, 0), Color(.5, .5, .6, 0), Color(.5, .5, .5, 0)));

    AviSaver avi;                                                            // main.cpp:76-81
    avi.Open(prefix + "anim.avi", px, py, 15, 1);
    std::vector<uint8_t> allData((size_t)px * py * 3);
    double time = 0;
    for (int count = 0; count < frames; time += 0.015, count++) {           // main.cpp:83
        if (scene->setTime(time)) { std::fprintf(stderr, "setTime failed: %s\n", scene->lastError()); return 1; }
        if (scene->runParallelly(11)) { std::fprintf(stderr, "render failed: %s\n", scene->lastError()); return 1; }
        char name[512];
        std::snprintf(name, sizeof name, "%s%03d.bmp", prefix.c_str(), count);
        if (scene->saveToBpm(name)) { std::fprintf(stderr, "saveToBpm failed: %s\n", scene->lastError()); return 1; }
        for (int i = 0; i < py; i++)                                         // main.cpp:89-98
            for (int j = 0; j < px; j++) {
                Color color;
                scene->queryColor(j, i, color);
                allData[((size_t)i * px + j) * 3 + 2] = (uint8_t)int(color.x_ * 255);
                allData[((size_t)i * px + j) * 3 + 1] = (uint8_t)int(color.y_ * 255);
                allData[((size_t)i * px + j) * 3 + 0] = (uint8_t)int(color.z_ * 255);
            }
        avi.AddImg(allData.data());
        std::printf("time %.3f -> %s (%llu rays)\n", time, name, (unsigned long long)scene->raysTraced());
    }
    avi.Save();
    return 0;
}
