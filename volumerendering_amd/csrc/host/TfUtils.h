// Mirrors med::TfUtils::CheckDragBounds (App/src/tf/TfUtils.cpp:9-43): keeps a dragged control point inside
// the plot and at least one texel away from its neighbours; the first and last point cannot move in x.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

#include "vrm.h"

namespace med {
class TfUtils {
public:
    static void CheckDragBounds(int index, std::vector<vrm::dvec2>& controlPoints, int maxDataVal)
    {
        const double X_MAX = maxDataVal - 1.0, X_MIN = 0.0;
        auto& cp = controlPoints[index];
        const int n = static_cast<int>(controlPoints.size());
        if (index == 0) cp.x = X_MIN;
        else if (index == n - 1) cp.x = X_MAX;
        cp.y = std::clamp(cp.y, 0.0, 1.0);
        cp.x = std::clamp(cp.x, X_MIN, X_MAX);
        if (index - 1 >= 0 && controlPoints[index - 1].x >= (cp.x - 1.0)) cp.x = std::ceil(controlPoints[index - 1].x + 1.0);
        if (index + 1 < n && controlPoints[index + 1].x <= (cp.x + 1.0)) cp.x = std::floor(controlPoints[index + 1].x - 1.0);
    }
};
}  // namespace med
