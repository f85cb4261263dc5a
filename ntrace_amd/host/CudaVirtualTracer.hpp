// CudaVirtualTracer.hpp -- tracer plugin interface (src/rt/cuda/CudaVirtualTracer.hpp:11-26).
#pragma once
#include "CudaAS.hpp"
#include "Scene.hpp"

namespace FW {

class Window;  // GUI message window: not part of this backend, kept for signature parity

class CudaVirtualTracer {
public:
    virtual ~CudaVirtualTracer(void) {}
    virtual void      setMessageWindow(Window* window) = 0;
    virtual void      setKernel(const String& kernelName) = 0;
    virtual BVHLayout getDesiredBVHLayout(void) const = 0;
    virtual void      setBVH(CudaAS* as) = 0;
    void              setScene(Scene* scene) { m_scene = scene; }
    virtual F32       traceBatch(RayBuffer& rays) = 0;  // returns launch time in seconds

protected:
    Scene* m_scene;
};

}  // namespace FW
