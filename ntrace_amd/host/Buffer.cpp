// Buffer.cpp -- see Buffer.hpp.  Behaviour follows src/framework/gpu/Buffer.cpp:238-
// (setOwner's valid/dirty state machine) for the CPU and device modules.
#include "Buffer.hpp"

#include <cstdlib>
#include <istream>
#include <ostream>

#include "ntrace_amd.h"

namespace FW {

static void check(int rc, const char* what)
{
    if (rc != NTR_OK) fail("Buffer: %s failed: %s", what, ntr_last_error());
}

void Buffer::init(S64 size)
{
    if (size < 0) fail("Buffer: negative size");
    m_size = size;
    m_original = Module_None;
    m_owner = Module_None;
    m_exists = Module_None;
    m_dirty = Module_None;
    m_cpuPtr = NULL;
    m_cudaPtr = NULL;
}

void Buffer::deinit(void)
{
    if ((m_exists & CPU) && m_original != CPU) ::free(m_cpuPtr);
    if ((m_exists & Cuda) && m_original != Cuda) ntr_free(m_cudaPtr);
    m_exists = Module_None;
    m_cpuPtr = NULL;
    m_cudaPtr = NULL;
}

void Buffer::wrapCPU(void* cpuPtr, S64 size)
{
    deinit();
    init(size);
    m_cpuPtr = (U8*)cpuPtr;
    m_original = m_owner = CPU;
    m_exists = CPU;
}

void Buffer::wrapCuda(CUdeviceptr cudaPtr, S64 size)
{
    deinit();
    init(size);
    m_cudaPtr = cudaPtr;
    m_original = m_owner = Cuda;
    m_exists = Cuda;
}

void Buffer::free(Module module)
{
    if ((m_exists & module) == 0 || m_original == module || m_exists == (U32)module) return;
    setOwner(module == CPU ? Cuda : CPU, false);
    if (module == CPU) { ::free(m_cpuPtr); m_cpuPtr = NULL; }
    else { ntr_free(m_cudaPtr); m_cudaPtr = NULL; }
    m_exists &= ~module;
    m_dirty &= ~module;
}

void Buffer::realloc(S64 size)
{
    if (size == m_size) return;
    if (m_original != Module_None) fail("Buffer: cannot resize a wrapped buffer");
    // Keep the contents: migrate through the current owner.
    Buffer tmp;
    tmp.m_size = size;
    S64 keep = FW::min(size, m_size);
    if (keep > 0 && m_owner != Module_None) {
        if (m_owner == Cuda) {
            tmp.setOwner(Cuda, true);
            check(ntr_memcpy_d2d(tmp.m_cudaPtr, m_cudaPtr, (size_t)keep, NULL), "ntr_memcpy_d2d");
            check(ntr_stream_synchronize(NULL), "ntr_stream_synchronize");
        } else {
            tmp.setOwner(CPU, true);
            memcpy(tmp.m_cpuPtr, m_cpuPtr, (size_t)keep);
        }
    }
    deinit();
    m_size = tmp.m_size;
    m_owner = tmp.m_owner;
    m_exists = tmp.m_exists;
    m_dirty = tmp.m_dirty;
    m_cpuPtr = tmp.m_cpuPtr;
    m_cudaPtr = tmp.m_cudaPtr;
    tmp.m_exists = Module_None;  // ownership moved
    tmp.m_cpuPtr = NULL;
    tmp.m_cudaPtr = NULL;
}

void Buffer::validateCPU(void)
{
    if ((m_exists & CPU) == 0 || (m_dirty & CPU) == 0) return;
    if ((m_exists & Cuda) != 0 && (m_dirty & Cuda) == 0 && m_size)
        check(ntr_memcpy_d2h(m_cpuPtr, m_cudaPtr, (size_t)m_size, NULL), "ntr_memcpy_d2h");
    m_dirty &= ~CPU;
}

void Buffer::setOwner(Module module, bool modify)
{
    if (m_owner == module) {
        if (modify) m_dirty = Module_All - module;
        return;
    }
    if (module == CPU) {
        if ((m_exists & CPU) == 0) {
            m_cpuPtr = (U8*)::malloc((size_t)FW::max(m_size, (S64)1));
            if (!m_cpuPtr) fail("Buffer: out of host memory");
            m_exists |= CPU;
            m_dirty |= CPU;
        }
        validateCPU();
    }
    if (module == Cuda) {
        if ((m_exists & Cuda) == 0) {
            check(ntr_malloc(&m_cudaPtr, (size_t)FW::max(m_size, (S64)1)), "ntr_malloc");
            m_exists |= Cuda;
            m_dirty |= Cuda;
        }
        if ((m_dirty & Cuda) != 0) {
            validateCPU();
            if ((m_exists & CPU) != 0 && m_size)
                check(ntr_memcpy_h2d(m_cudaPtr, m_cpuPtr, (size_t)m_size, NULL), "ntr_memcpy_h2d");
            m_dirty &= ~Cuda;
        }
    }
    m_owner = module;
    if (modify) m_dirty = Module_All - module;
}

void Buffer::getRange(void* dst, S64 srcOfs, S64 size) const
{
    if (!size) return;
    Buffer* self = const_cast<Buffer*>(this);
    if (m_owner == Cuda)
        check(ntr_memcpy_d2h(dst, (const U8*)self->getCudaPtr() + srcOfs, (size_t)size, NULL), "ntr_memcpy_d2h");
    else
        memcpy(dst, self->getPtr(srcOfs), (size_t)size);
}

void Buffer::setRange(S64 dstOfs, const void* src, S64 size)
{
    if (!size) return;
    if (m_owner == Cuda)
        check(ntr_memcpy_h2d((U8*)getMutableCudaPtr() + dstOfs, src, (size_t)size, NULL), "ntr_memcpy_h2d");
    else
        memcpy(getMutablePtr(dstOfs), src, (size_t)size);
}

void Buffer::setRange(S64 dstOfs, Buffer& src, S64 srcOfs, S64 size)
{
    if (!size) return;
    if (src.m_owner == Cuda && (m_owner == Cuda || m_owner == Module_None)) {
        check(ntr_memcpy_d2d((U8*)getMutableCudaPtr() + dstOfs, (const U8*)src.getCudaPtr() + srcOfs, (size_t)size, NULL),
              "ntr_memcpy_d2d");
        check(ntr_stream_synchronize(NULL), "ntr_stream_synchronize");
    } else
        setRange(dstOfs, src.getPtr(srcOfs), size);
}

void Buffer::clearRange(S64 dstOfs, int value, S64 size)
{
    if (!size) return;
    if (m_owner == Cuda) {
        check(ntr_memset((U8*)getMutableCudaPtr() + dstOfs, value, (size_t)size, NULL), "ntr_memset");
        check(ntr_stream_synchronize(NULL), "ntr_stream_synchronize");
    } else
        memset(getMutablePtr(dstOfs), value, (size_t)size);
}

void Buffer::readFromStream(std::istream& s)
{
    S64 size = 0;
    s.read((char*)&size, sizeof(size));
    if (!s || size < 0) { setError("Buffer: truncated stream"); return; }
    resizeDiscard(size);
    if (size) s.read((char*)getMutablePtrDiscard(), (std::streamsize)size);
    if (!s) setError("Buffer: truncated stream");
}

void Buffer::writeToStream(std::ostream& s)
{
    S64 size = m_size;
    s.write((const char*)&size, sizeof(size));
    if (size) s.write((const char*)getPtr(), (std::streamsize)size);
}

}  // namespace FW
