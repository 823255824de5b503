// stand-in for boost::alignment::aligned_allocator (tools/refprobe/README.md): posix_memalign-backed std allocator.  Written for this
// repository; contains no Boost text.
#pragma once
#include <cstddef>
#include <cstdlib>
#include <new>
namespace boost { namespace alignment {
template <class T, std::size_t Alignment>
struct aligned_allocator {
    typedef T value_type;
    typedef T *pointer;
    typedef const T *const_pointer;
    typedef T &reference;
    typedef const T &const_reference;
    typedef std::size_t size_type;
    typedef std::ptrdiff_t difference_type;
    template <class U> struct rebind { typedef aligned_allocator<U, Alignment> other; };
    aligned_allocator() {}
    template <class U> aligned_allocator(const aligned_allocator<U, Alignment> &) {}
    pointer allocate(size_type n, const void * = 0)
    {
        void *p = 0;
        const std::size_t al = Alignment < sizeof(void *) ? sizeof(void *) : Alignment;
        if (posix_memalign(&p, al, n ? n * sizeof(T) : al) != 0) throw std::bad_alloc();
        return static_cast<pointer>(p);
    }
    void deallocate(pointer p, size_type) { std::free(p); }
    size_type max_size() const { return static_cast<size_type>(-1) / sizeof(T); }
    void construct(pointer p, const T &v) { new (static_cast<void *>(p)) T(v); }
    void destroy(pointer p) { p->~T(); }
};
template <class A, class B, std::size_t N> bool operator==(const aligned_allocator<A, N> &, const aligned_allocator<B, N> &) { return true; }
template <class A, class B, std::size_t N> bool operator!=(const aligned_allocator<A, N> &, const aligned_allocator<B, N> &) { return false; }
} }
