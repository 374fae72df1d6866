// Camera -- orbit camera (pitch / yaw / distance about a focal point) producing the view / projection matrices
// and their inverses.  Mirrors med::Camera (App/src/Camera.h:15-76).
#pragma once
#include "vrm.h"

namespace med {

enum class CameraType { Undefined, Perspective, Orthographic };

class Camera {
    Camera(float fov, float aspect, float near, float far);
    Camera(float left, float right, float bottom, float top, float near, float far);

public:
    static Camera CreatePerspective(float fov, float aspect, float near, float far);
    static Camera CreateOrthographic(float left, float right, float bottom, float top, float near, float far);

    enum Key { KeyUp, KeyDown, KeyLeft, KeyRight };  // stands in for the GLFW arrow-key codes
    void KeyboardEvent(int key);

    void SetPosition(const vrm::vec3 position);
    void SetAspectRatio(float aspectRatio);
    void SetFov(float fov);
    void SetZoomDistance(float delta);
    void Rotate(float delta_x, float delta_y);

    const vrm::mat4& GetProjectionMatrix() const { return m_ProjectionMatrix; }
    const vrm::mat4& GetViewMatrix() const { return m_ViewMatrix; }
    const vrm::mat4& GetInverseProjectionMatrix() const { return m_InverseProjectionMatrix; }
    const vrm::mat4& GetInverseViewMatrix() const { return m_InverseViewMatrix; }
    vrm::quat GetOrientation() const;
    vrm::vec3 GetUp() const;
    vrm::vec3 GetForward() const;
    vrm::vec3 GetRight() const;
    vrm::vec3 GetPosition() const;
    float GetZoom() const { return m_Distance; }

    // direct setters for scripted (non-interactive) use: pitch / yaw in radians, orbit distance
    void SetOrbit(float pitch, float yaw, float distance);

private:
    void RecalculateViewMatrix();
    void RecalculateProjectionMatrix();

    CameraType m_Type = CameraType::Undefined;
    float m_Far = 0.0f, m_Near = 0.0f;
    float m_Fov = 0.0f, m_Aspect = 0.0f;
    float m_Left = 0.0f, m_Right = 0.0f, m_Bottom = 0.0f, m_Top = 0.0f;
    float m_Speed = 0.1f, m_ZoomSpeed = 0.01f;
    float m_Pitch = 0.0f, m_Yaw = 0.0f;
    float m_RotateSens = 0.005f;
    float m_Distance = 5.0f;
    vrm::mat4 m_ProjectionMatrix{1.0f};
    vrm::vec3 m_Position{0.0f, 0.0f, 0.0f};
    vrm::mat4 m_ViewMatrix{1.0f};
    vrm::mat4 m_InverseViewMatrix{1.0f};
    vrm::mat4 m_InverseProjectionMatrix{1.0f};
};

}  // namespace med
