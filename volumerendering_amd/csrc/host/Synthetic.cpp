// Synthetic.cpp -- deterministic synthetic inputs of SURVEY.md section 8d (no DICOM data exists offline),
// generated multi-threaded for the full-size benchmark volumes.  Same definitions (f64 arithmetic, value for
// value) as volumerendering_amd/synth.py, which the small parity cases use; tests compare the two.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <functional>
#include <thread>
#include <vector>

namespace {
void par_z(int n, const std::function<void(int, int)>& fn)
{
    unsigned w = std::max(1u, std::thread::hardware_concurrency());
    w = std::min<unsigned>(w, (unsigned)n);
    std::vector<std::thread> th;
    int chunk = (n + (int)w - 1) / (int)w;
    for (unsigned t = 0; t < w; ++t) {
        int b = std::min(n, (int)t * chunk), e = std::min(n, b + chunk);
        if (b < e) th.emplace_back(fn, b, e);
    }
    for (auto& t : th) t.join();
}
inline uint32_t xorshift32(uint32_t x)
{
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    return x;
}
inline double sq(double v) { return v * v; }
}  // namespace

extern "C" {

// nested ellipsoids: air 0 / soft tissue 1000-1100 / bone shell 2500-3000 / interior 1040, + noise in [-40, 40]
// from xorshift32(0x5EED ^ voxel index) inside the body; 12-bit values.  air_noise != 0: the air outside the body
// carries the same generator's noise as raw 0..80 -- what a scanner delivers (stored CT values of air are not an
// exact constant), so that nothing about the volume is exactly zero.
void vrh_synth_ct_phantom_air(int n, int air_noise, uint16_t* out)
{
    const double c = (n - 1) / 2.0, h = n / 2.0;
    par_z(n, [&](int z0, int z1) {
        for (int z = z0; z < z1; ++z)
            for (int y = 0; y < n; ++y)
                for (int x = 0; x < n; ++x) {
                    const double qx = (x - c) / h, qy = (y - c) / h, qz = (z - c) / h;
                    const bool body = sq(qx / 0.85) + sq(qy / 0.70) + sq(qz / 0.90) < 1.0;
                    const bool outer = sq(qx / 0.55) + sq(qy / 0.45) + sq(qz / 0.60) < 1.0;
                    const bool inner = sq(qx / 0.45) + sq(qy / 0.35) + sq(qz / 0.50) < 1.0;
                    double val = 0.0;
                    if (body) val = 1000.0 + 100.0 * (0.5 + 0.5 * qz);
                    if (outer && !inner) val = 2500.0 + 500.0 * (0.5 + 0.5 * qx);
                    if (inner) val = 1040.0;
                    const uint32_t idx = (uint32_t)(((uint64_t)z * n + y) * n + x);
                    const double noise = (double)((int64_t)(xorshift32(0x5EEDu ^ idx) % 81u) - 40);
                    val = body ? val + noise : (air_noise ? noise + 40.0 : 0.0);
                    val = std::nearbyint(val);
                    out[((size_t)z * n + y) * n + x] = (uint16_t)std::min(4095.0, std::max(0.0, val));
                }
    });
}

void vrh_synth_ct_phantom(int n, uint16_t* out) { vrh_synth_ct_phantom_air(n, 0, out); }

// sphere-N: round(4095 * max(0, 1 - |p - c| / (0.45 N)))
void vrh_synth_sphere(int n, uint16_t* out)
{
    const double c = (n - 1) / 2.0;
    par_z(n, [&](int z0, int z1) {
        for (int z = z0; z < z1; ++z)
            for (int y = 0; y < n; ++y)
                for (int x = 0; x < n; ++x) {
                    const double r = std::sqrt(sq(x - c) + sq(y - c) + sq(z - c));
                    out[((size_t)z * n + y) * n + x] = (uint16_t)std::nearbyint(4095.0 * std::max(0.0, 1.0 - r / (0.45 * n)));
                }
    });
}

// n^3 vec4 mask: channel r = ellipsoid A, g = ellipsoid B, exactly 0 / 1
void vrh_synth_mask(int n, float* out)
{
    const double c = (n - 1) / 2.0, h = n / 2.0;
    par_z(n, [&](int z0, int z1) {
        for (int z = z0; z < z1; ++z)
            for (int y = 0; y < n; ++y)
                for (int x = 0; x < n; ++x) {
                    const double qx = (x - c) / h, qy = (y - c) / h, qz = (z - c) / h;
                    const bool a = sq((qx - 0.15) / 0.25) + sq((qy + 0.05) / 0.20) + sq(qz / 0.30) < 1.0;
                    const bool b = sq((qx + 0.25) / 0.15) + sq((qy - 0.10) / 0.15) + sq((qz + 0.1) / 0.20) < 1.0;
                    float* o = out + 4 * (((size_t)z * n + y) * n + x);
                    o[0] = a ? 1.0f : 0.0f;
                    o[1] = b ? 1.0f : 0.0f;
                    o[2] = 0.0f;
                    o[3] = 0.0f;
                }
    });
}

}  // extern "C"
