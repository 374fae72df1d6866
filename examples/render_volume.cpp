// render_volume.cpp -- the reference's start-up + one frame of its main loop, C++ only, on the MI355X ray-marcher:
//   Application::Application -> OnStart(scene) -> OnUpdate -> OnRender -> present      (App/src/Application.cpp:39-239)
// with the re-implemented host classes of csrc/host/ above the C ABI of include/vr.h.  Renders BASELINE config C1
// (sphere-64, 256x256, BasicVolumeApp) or, with an argument, a sphere of that size with the lit shader, writes the
// presented frame as frame.ppm and prints the composited-sample count.
//
//   g++ -std=c++20 -O2 -I include -I volumerendering_amd/csrc/host examples/render_volume.cpp \
//       -L volumerendering_amd -lvr_host -lvr_hip -Wl,-rpath,$PWD/volumerendering_amd -o render_volume
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "Application.h"

int main(int argc, char** argv)
{
    const int n = argc > 1 ? std::atoi(argv[1]) : 64;
    const bool lit = argc > 1;
    const uint32_t W = 256, H = 256;
    if (n < 2 || n > 1024) return 1;

    // sphere-N of SURVEY.md 8d: raw = round(4095 * max(0, 1 - |p - c| / (0.45 N)))
    std::vector<uint16_t> raw((size_t)n * n * n);
    const double c = (n - 1) / 2.0;
    for (int z = 0; z < n; ++z)
        for (int y = 0; y < n; ++y)
            for (int x = 0; x < n; ++x) {
                const double r = std::sqrt((x - c) * (x - c) + (y - c) * (y - c) + (z - c) * (z - c));
                raw[((size_t)z * n + y) * n + x] = (uint16_t)std::nearbyint(4095.0 * std::max(0.0, 1.0 - r / (0.45 * n)));
            }
    auto ct = std::make_shared<med::VolumeFile>(
        med::VolumeFile::FromRaw(raw.data(), {(uint16_t)n, (uint16_t)n, (uint16_t)n}));

    med::Application app(W, H, /*device*/ 0);
    if (!app.Ok()) {
        std::fprintf(stderr, "no usable MI355X: %s\n", app.LastError().c_str());
        return 2;
    }
    int rc = lit ? app.OnStart(std::make_unique<med::BasicVolLightApp>(ct)) : app.OnStart(std::make_unique<med::BasicVolumeApp>(ct));
    if (rc != VR_OK) {
        std::fprintf(stderr, "scene start failed: %s\n", app.LastError().c_str());
        return 3;
    }
    app.GetCamera().SetOrbit(0.35f, 0.6f, 1.2f);   // pitch, yaw, distance
    if (app.OnUpdate() != VR_OK || app.OnRender() != VR_OK) {
        std::fprintf(stderr, "render failed: %s\n", app.LastError().c_str());
        return 4;
    }
    std::vector<float> frag((size_t)W * H * 4);
    std::vector<uint8_t> bgra((size_t)W * H * 4);
    uint64_t samples = 0;
    if (app.ReadFrame(frag.data(), bgra.data(), &samples) != VR_OK) return 5;

    if (FILE* f = std::fopen("frame.ppm", "wb")) {
        std::fprintf(f, "P6\n%u %u\n255\n", W, H);
        for (size_t i = 0; i < (size_t)W * H; ++i) {
            const unsigned char rgb[3] = {bgra[4 * i + 2], bgra[4 * i + 1], bgra[4 * i + 0]};
            std::fwrite(rgb, 1, 3, f);
        }
        std::fclose(f);
    }
    double sum = 0.0;
    for (float v : frag) sum += v;
    std::printf("%s %d^3 %ux%u: %llu composited samples, sum of the fragment outputs %.15g\n", lit ? "lit" : "unlit", n, W, H,
                (unsigned long long)samples, sum);
    return 0;
}
