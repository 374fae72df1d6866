// Linear ramp generator used by both transfer functions.  Mirrors med::LinearInterpolation::Generate
// (App/src/tf/LinearInterpolation.h:10-33): value(i) = f(x0) + slope * (i - x0) with
// slope = (f(x1) - f(x0)) * (1.0f / float(x1 - x0)), for i = x0 .. x1 inclusive; the vec4 form interpolates
// rgb and forces alpha to 1.
#pragma once
#include <vector>

#include "vrm.h"

namespace med {

class LinearInterpolation {
public:
    static std::vector<float> Generate(int x0, int x1, float fx0, float fx1, int step = 1)
    {
        std::vector<float> out;
        const float slope = (fx1 - fx0) * (1.0f / static_cast<float>(x1 - x0));
        for (int i = x0; i < x1 + 1; i += step) out.push_back(fx0 + slope * (i - x0));
        return out;
    }

    static std::vector<vrm::vec4> Generate(int x0, int x1, vrm::vec4 fx0, vrm::vec4 fx1, int step = 1)
    {
        std::vector<vrm::vec4> out;
        const vrm::vec4 slope = (fx1 - fx0) * (1.0f / static_cast<float>(x1 - x0));
        for (int i = x0; i < x1 + 1; i += step) {
            float r = fx0.r + slope.r * (i - x0);
            float g = fx0.g + slope.g * (i - x0);
            float b = fx0.b + slope.b * (i - x0);
            out.emplace_back(r, g, b, 1.0f);
        }
        return out;
    }
};

}  // namespace med
