// Camera -- follows App/src/Camera.cpp: view = inverse(T(-forward*distance) * T(position) * R(orientation))
// (:154-162), projection = perspective / ortho and its inverse (:164-177), zoom clamp 0.1..10 (:52-70).
#include "Camera.h"

namespace med {

Camera::Camera(float left, float right, float bottom, float top, float near, float far)
    : m_Type(CameraType::Orthographic), m_Far(far), m_Near(near), m_Left(left), m_Right(right), m_Bottom(bottom), m_Top(top)
{
    RecalculateViewMatrix();
    RecalculateProjectionMatrix();
}

Camera::Camera(float fov, float aspect, float near, float far)
    : m_Type(CameraType::Perspective), m_Far(far), m_Near(near), m_Fov(fov), m_Aspect(aspect)
{
    RecalculateViewMatrix();
    RecalculateProjectionMatrix();
}

Camera Camera::CreatePerspective(float fov, float aspect, float near, float far) { return {fov, aspect, near, far}; }
Camera Camera::CreateOrthographic(float l, float r, float b, float t, float n, float f) { return {l, r, b, t, n, f}; }

void Camera::SetPosition(const vrm::vec3 position)
{
    m_Position = position;
    RecalculateViewMatrix();
}

void Camera::SetAspectRatio(float aspectRatio)
{
    m_Aspect = aspectRatio;
    RecalculateProjectionMatrix();
}

void Camera::SetFov(float fov)
{
    m_Fov = fov;
    RecalculateProjectionMatrix();
}

void Camera::SetZoomDistance(float delta)
{
    if (m_Type == CameraType::Perspective) {
        m_Distance = vrm::clamp(m_Distance + delta * m_ZoomSpeed, 0.1f, 10.0f);
        RecalculateViewMatrix();
    } else {
        const float sign = (delta < 0.0f) ? -1.0f : 1.0f;
        const float aspect = m_Right / m_Top;
        m_Left -= (m_ZoomSpeed * sign) * aspect;
        m_Right += (m_ZoomSpeed * sign) * aspect;
        m_Bottom -= m_ZoomSpeed * sign;
        m_Top += m_ZoomSpeed * sign;
        RecalculateProjectionMatrix();
    }
}

void Camera::SetOrbit(float pitch, float yaw, float distance)
{
    m_Pitch = pitch;
    m_Yaw = yaw;
    m_Distance = distance;
    RecalculateViewMatrix();
}

vrm::quat Camera::GetOrientation() const { return vrm::quat_from_euler(vrm::vec3(m_Pitch, m_Yaw, 0.0f)); }
vrm::vec3 Camera::GetUp() const { return vrm::normalize(vrm::rotate(GetOrientation(), vrm::vec3(0.0f, 1.0f, 0.0f))); }
vrm::vec3 Camera::GetForward() const { return vrm::normalize(vrm::rotate(GetOrientation(), vrm::vec3(0.0f, 0.0f, -1.0f))); }
vrm::vec3 Camera::GetRight() const { return vrm::normalize(vrm::rotate(GetOrientation(), vrm::vec3(1.0f, 0.0f, 0.0f))); }
vrm::vec3 Camera::GetPosition() const { return m_Position - GetForward() * m_Distance; }

void Camera::KeyboardEvent(int key)
{
    switch (key) {
    case KeyUp: m_Position = m_Position + GetForward() * m_Speed; break;
    case KeyDown: m_Position = m_Position + (-GetForward()) * m_Speed; break;
    case KeyLeft: m_Position = m_Position + (-GetRight()) * m_Speed; break;
    case KeyRight: m_Position = m_Position + GetRight() * m_Speed; break;
    default: break;
    }
    RecalculateViewMatrix();
}

void Camera::Rotate(float delta_x, float delta_y)
{
    m_Pitch += delta_y * m_RotateSens;
    m_Yaw += delta_x * m_RotateSens;
    RecalculateViewMatrix();
}

void Camera::RecalculateViewMatrix()
{
    const vrm::quat orientation = GetOrientation();
    const vrm::mat4 transform = vrm::translate(vrm::mat4(1.0f), -GetForward() * m_Distance) *
                                vrm::translate(vrm::mat4(1.0f), m_Position) * vrm::to_mat4(orientation);
    m_ViewMatrix = vrm::inverse(transform);
    m_InverseViewMatrix = transform;
}

void Camera::RecalculateProjectionMatrix()
{
    m_ProjectionMatrix = (m_Type == CameraType::Perspective) ? vrm::perspective(m_Fov, m_Aspect, m_Near, m_Far)
                                                             : vrm::ortho(m_Left, m_Right, m_Bottom, m_Top, m_Near, m_Far);
    m_InverseProjectionMatrix = vrm::inverse(m_ProjectionMatrix);
}

}  // namespace med
