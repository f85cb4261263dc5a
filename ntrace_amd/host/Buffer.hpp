// Buffer.hpp -- HIP-backed mirror of the reference's FW::Buffer
// (src/framework/gpu/Buffer.hpp:38-113): a byte buffer that exists on the CPU
// and/or the device, with an owner and per-module dirty bits and lazy migration.
//
// Same public semantics for the subset the tracer path uses:
//   getPtr()            makes the CPU copy valid (may copy D->H)
//   getMutablePtr()     same, and marks the device copy dirty
//   getCudaPtr()        makes the device copy valid (may copy H->D)   [HIP pointer]
//   getMutableCudaPtr() same, and marks the CPU copy dirty
//   wrapCPU/wrapCuda    borrow external memory (never freed here)
// GL interop, page-lock hints and async copies are not part of this backend.
// Device memory goes through the C-ABI (ntr_malloc / ntr_memcpy_*), so this class
// contains no HIP types.
#pragma once
#include <iosfwd>

#include "Defs.hpp"

namespace FW {

typedef void* CUdeviceptr;  // the reference name is kept; the value is a HIP device pointer

class Buffer {
public:
    enum Module { CPU = 1 << 0, Cuda = 1 << 2, Module_None = 0, Module_All = CPU | Cuda };

    explicit Buffer(void) { init(0); }
    explicit Buffer(const void* ptr, S64 size) { init(size); if (ptr) set(ptr); }
    Buffer(Buffer& other) { init(other.getSize()); setRange(0, other, 0, other.getSize()); }
    ~Buffer(void) { deinit(); }

    void wrapCPU(void* cpuPtr, S64 size);
    void wrapCuda(CUdeviceptr cudaPtr, S64 size);

    S64  getSize(void) const { return m_size; }
    void reset(void) { deinit(); init(0); }
    void reset(const void* ptr, S64 size) { deinit(); init(size); if (ptr) setRange(0, ptr, size); }
    void resize(S64 size) { realloc(size); }                                  // keeps contents
    void resizeDiscard(S64 size) { if (m_size != size) reset(NULL, size); }  // drops contents
    void free(Module module);

    void getRange(void* dst, S64 srcOfs, S64 size) const;
    void get(void* ptr) { getRange(ptr, 0, getSize()); }
    void setRange(S64 dstOfs, const void* src, S64 size);
    void setRange(S64 dstOfs, Buffer& src, S64 srcOfs, S64 size);
    void set(const void* ptr) { setRange(0, ptr, getSize()); }
    void set(const void* ptr, S64 size) { resizeDiscard(size); setRange(0, ptr, size); }
    void set(Buffer& other) { if (&other != this) { resizeDiscard(other.getSize()); setRange(0, other, 0, other.getSize()); } }
    void clearRange(S64 dstOfs, int value, S64 size);
    void clear(int value = 0) { clearRange(0, value, m_size); }

    void   setOwner(Module module, bool modify);
    Module getOwner(void) const { return m_owner; }
    void   discard(void) { m_dirty = 0; }

    const U8*   getPtr(S64 ofs = 0) { setOwner(CPU, false); return m_cpuPtr + ofs; }
    U8*         getMutablePtr(S64 ofs = 0) { setOwner(CPU, true); return m_cpuPtr + ofs; }
    U8*         getMutablePtrDiscard(S64 ofs = 0) { discard(); return getMutablePtr(ofs); }
    CUdeviceptr getCudaPtr(S64 ofs = 0) { setOwner(Cuda, false); return (U8*)m_cudaPtr + ofs; }
    CUdeviceptr getMutableCudaPtr(S64 ofs = 0) { setOwner(Cuda, true); return (U8*)m_cudaPtr + ofs; }
    CUdeviceptr getMutableCudaPtrDiscard(S64 ofs = 0) { discard(); return getMutableCudaPtr(ofs); }

    Buffer& operator=(Buffer& other) { set(other); return *this; }

    // Serializable (src/framework/gpu/Buffer.cpp:349-381): S64 size + raw bytes, little endian.
    void readFromStream(std::istream& s);
    void writeToStream(std::ostream& s);

private:
    void init(S64 size);
    void deinit(void);
    void realloc(S64 size);
    void validateCPU(void);

    S64    m_size;
    Module m_original;  // module that wraps external memory, if any
    Module m_owner;
    U32    m_exists;
    U32    m_dirty;
    U8*    m_cpuPtr;
    void*  m_cudaPtr;
};

}  // namespace FW
