// CameraControls.hpp -- the camera state NTrace benchmarks are specified with
// (src/framework/3d/CameraControls.hpp/.cpp): the signature codec (:342-399, 471-545), the
// world->camera / perspective matrices (:251-284, base/Math.cpp:79-92) and the primary-ray matrix
// invert(fitToView(-1, 2) * worldToClip) of Renderer::beginFrame (Renderer.cpp:473-477).
// GUI interaction is out of scope.
#pragma once
#include "RayGen.hpp"

namespace FW {

class CameraControls {
public:
    CameraControls(void);

    String encodeSignature(void) const;
    void   decodeSignature(const String& sig);  // sets the sticky error on malformed input

    const Vec3f& getPosition(void) const { return m_position; }
    const Vec3f& getForward(void) const { return m_forward; }
    const Vec3f& getUp(void) const { return m_up; }
    F32 getSpeed(void) const { return m_speed; }
    F32 getFOV(void) const { return m_fov; }
    F32 getNear(void) const { return m_near; }
    F32 getFar(void) const { return m_far; }
    bool getKeepAligned(void) const { return m_keepAligned; }
    void setPosition(const Vec3f& v) { m_position = v; }
    void setForward(const Vec3f& v) { m_forward = v; }
    void setUp(const Vec3f& v) { m_up = v; }
    void setFOV(F32 v) { m_fov = v; }
    void setNear(F32 v) { m_near = v; }
    void setFar(F32 v) { m_far = v; }

    Mat4f getWorldToCamera(void) const;
    Mat4f getCameraToClip(void) const;  // Mat4f::perspective(fov, near, far)
    Mat4f getWorldToClip(void) const;
    // invert(Mat4f::fitToView(-1, 2, viewSize) * worldToClip): nscreen -> world for rayGenPrimaryKernel
    Mat4f getNScreenToWorld(S32 viewW, S32 viewH) const;
    CameraView getView(S32 viewW, S32 viewH) const;

private:
    Vec3f m_position, m_forward, m_up;
    F32   m_speed, m_fov, m_near, m_far;
    bool  m_keepAligned;
};

Mat4f mat4Mul(const Mat4f& a, const Mat4f& b);
Mat4f mat4Inverted(const Mat4f& a);  // cofactor inverse, MatrixBase::inverted (Math.hpp:1024-1046)

}  // namespace FW
