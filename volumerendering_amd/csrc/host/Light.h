// Mirrors med::Light (App/src/renderer/Light.h:9-14): three vec4 (rgb used, w pads to 16 B).
#pragma once
#include "vrm.h"
namespace med {
struct Light {
    vrm::vec4 Position{0.0f};
    vrm::vec4 Ambient{0.0f};
    vrm::vec4 Diffuse{0.0f};
};
}  // namespace med
