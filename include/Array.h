/*
 * Array<T,N> (host) and Array_gpu<T,N> (device) with the interface of /root/reference/include/Array.h:
 * column-major, 1-based operator()({i,j,..}), dim(i) 1-based, set_offsets, subset, fill, ptr, v, dump.
 * Array_gpu owns HBM obtained through the C ABI (rrx_malloc); copy = deep D2D copy, move steals, construction
 * from an Array uploads, constructing an Array from an Array_gpu downloads (Array.h:367-623 of the reference).
 */
#ifndef ARRAY_H
#define ARRAY_H
#include <array>
#include <vector>
#include <string>
#include <fstream>
#include <algorithm>
#include <stdexcept>
#include "rrx_forward.h"

template<int N> inline std::array<int,N> calc_strides(const std::array<int,N>& dims)
{
    std::array<int,N> s; s[0] = 1;
    for (int i=1; i<N; ++i) s[i] = s[i-1]*dims[i-1];
    return s;
}
template<int N> inline int product(const std::array<int,N>& a) { int p = 1; for (int v : a) p *= v; return p; }

template<typename T, int N> class Array_gpu;

template<typename T, int N>
class Array
{
    public:
        Array() : dims({}), ncells(0), strides({}), offsets({}) {}
        Array(const std::array<int,N>& dims) : dims(dims), ncells(product<N>(dims)), data(ncells), strides(calc_strides<N>(dims)), offsets({}) {}
        Array(const std::vector<T>& d, const std::array<int,N>& dims) :
            dims(dims), ncells(product<N>(dims)), data(d.begin(), d.begin() + product<N>(dims)), strides(calc_strides<N>(dims)), offsets({}) {}
        Array(std::vector<T>&& d, const std::array<int,N>& dims) :
            dims(dims), ncells(product<N>(dims)), data(std::move(d)), strides(calc_strides<N>(dims)), offsets({}) { data.resize(ncells); }
        Array(const Array_gpu<T,N>& a);      // download

        void set_offsets(const std::array<int,N>& o) { offsets = o; }
        std::array<int,N> get_dims() const { return dims; }
        void set_dims(const std::array<int,N>& d)
        {
            if (ncells != 0) throw std::runtime_error("Only arrays of size 0 can be resized");
            dims = d; ncells = product<N>(d); data.resize(ncells); strides = calc_strides<N>(d); offsets = {};
        }
        std::vector<T>& v() { return data; }
        const std::vector<T>& v() const { return data; }
        T* ptr() { return data.data(); }
        const T* ptr() const { return data.data(); }
        int size() const { return ncells; }
        T max() const { return *std::max_element(data.begin(), data.end()); }
        T min() const { return *std::min_element(data.begin(), data.end()); }
        int dim(const int i) const { return dims[i-1]; }
        bool is_empty() const { return ncells == 0; }
        void fill(const T value) { std::fill(data.begin(), data.end(), value); }

        T& operator()(const std::array<int,N>& idx) { return data[index(idx)]; }
        T operator()(const std::array<int,N>& idx) const { return data[index(idx)]; }

        // ranges are 1-based inclusive {start, end} per dimension
        Array<T,N> subset(const std::array<std::array<int,2>,N>& ranges) const
        {
            std::array<int,N> sd;
            for (int i=0; i<N; ++i) sd[i] = ranges[i][1] - ranges[i][0] + 1;
            Array<T,N> out(sd);
            std::array<int,N> ix;
            for (int c=0; c<out.ncells; ++c)
            {
                int rem = c;
                for (int d=0; d<N; ++d) { ix[d] = (dims[d] == 1 ? 1 + offsets[d] : rem % sd[d] + ranges[d][0]); rem /= sd[d]; }
                out.data[c] = data[index(ix)];
            }
            return out;
        }

        void dump(const std::string& name) const
        {
            std::string file_name = name;
            for (int i=0; i<N; ++i) file_name += "." + std::to_string(dims[i]);
            file_name += ".bin";
            std::ofstream f(file_name, std::ios::out | std::ios::trunc | std::ios::binary);
            if (!f) throw std::runtime_error("cannot write file " + file_name);
            f.write(reinterpret_cast<const char*>(data.data()), size_t(ncells)*sizeof(T));
        }

    private:
        int index(const std::array<int,N>& idx) const
        {
            int s = 0;
            for (int i=0; i<N; ++i) s += (idx[i] - offsets[i] - 1) * strides[i];
            return s;
        }
        std::array<int,N> dims;
        int ncells;
        std::vector<T> data;
        std::array<int,N> strides;
        std::array<int,N> offsets;
        template<typename, int> friend class Array_gpu;
};


template<typename T, int N>
class Array_gpu
{
    public:
        Array_gpu() : dims({}), ncells(0), data_ptr(nullptr), strides({}), offsets({}) {}
        Array_gpu(const std::array<int,N>& dims) : dims(dims), ncells(product<N>(dims)), data_ptr(nullptr), strides(calc_strides<N>(dims)), offsets({}) { allocate(); }
        Array_gpu(const Array<T,N>& a) : dims(a.dims), ncells(a.ncells), data_ptr(nullptr), strides(a.strides), offsets(a.offsets)
        {
            allocate();
            if (ncells > 0) rrx_host::check(rrx_memcpy_h2d_stream(data_ptr, a.ptr(), size_t(ncells)*sizeof(T), rrx_host::current_stream()));
        }
        Array_gpu(const Array_gpu<T,N>& a) : dims(a.dims), ncells(a.ncells), data_ptr(nullptr), strides(a.strides), offsets(a.offsets)
        {
            allocate();
            if (ncells > 0) { rrx_host::check(rrx_memcpy_d2d(data_ptr, a.data_ptr, size_t(ncells)*sizeof(T), rrx_host::current_stream())); }
        }
        Array_gpu(Array_gpu<T,N>&& a) noexcept : dims(a.dims), ncells(a.ncells), data_ptr(a.data_ptr), strides(a.strides), offsets(a.offsets), owns(a.owns), alloc_stream(a.alloc_stream)
        { a.data_ptr = nullptr; a.ncells = 0; }
        // non-owning view of device memory managed elsewhere (reference: Array.h:476-486)
        Array_gpu(T* ptr, const std::array<int,N>& dims) : dims(dims), ncells(product<N>(dims)), data_ptr(ptr), strides(calc_strides<N>(dims)), offsets({}), owns(false) {}
        ~Array_gpu() { release(); }

        Array_gpu<T,N>& operator=(const Array_gpu<T,N>& a)
        {
            if (this == &a) return *this;
            if (ncells != a.ncells || !owns) { release(); ncells = a.ncells; owns = true; allocate(); }
            dims = a.dims; strides = a.strides; offsets = a.offsets;
            if (ncells > 0) rrx_host::check(rrx_memcpy_d2d(data_ptr, a.data_ptr, size_t(ncells)*sizeof(T), rrx_host::current_stream()));
            return *this;
        }
        Array_gpu<T,N>& operator=(Array_gpu<T,N>&& a) noexcept
        {
            if (this == &a) return *this;
            release();
            dims = a.dims; ncells = a.ncells; data_ptr = a.data_ptr; strides = a.strides; offsets = a.offsets; owns = a.owns;
            alloc_stream = a.alloc_stream;
            a.data_ptr = nullptr; a.ncells = 0;
            return *this;
        }
        Array_gpu<T,N>& operator=(const Array<T,N>& a) { *this = Array_gpu<T,N>(a); return *this; }

        void set_offsets(const std::array<int,N>& o) { offsets = o; }
        std::array<int,N> get_dims() const { return dims; }
        void set_dims(const std::array<int,N>& d)
        {
            if (ncells != 0) throw std::runtime_error("Only arrays of size 0 can be resized");
            dims = d; ncells = product<N>(d); strides = calc_strides<N>(d); offsets = {}; owns = true; allocate();
        }
        void set_data(const Array<T,N>& a)
        {
            if (a.size() != ncells) throw std::runtime_error("set_data: size mismatch");
            if (ncells > 0) rrx_host::check(rrx_memcpy_h2d_stream(data_ptr, a.ptr(), size_t(ncells)*sizeof(T), rrx_host::current_stream()));
        }
        void fill(const T value)
        {
            if (ncells == 0) return;
            std::vector<T> h(ncells, value);          // rarely used outside setup code
            rrx_host::check(rrx_memcpy_h2d_stream(data_ptr, h.data(), size_t(ncells)*sizeof(T), rrx_host::current_stream()));
        }
        T* ptr() { return data_ptr; }
        const T* ptr() const { return data_ptr; }
        int size() const { return ncells; }
        int dim(const int i) const { return dims[i-1]; }

        // synchronous single-element read-back, as in the reference (Array.h:567-573)
        T operator()(const std::array<int,N>& idx) const
        {
            int s = 0;
            for (int i=0; i<N; ++i) s += (idx[i] - offsets[i] - 1) * strides[i];
            T v;
            rrx_host::check(rrx_memcpy_d2h_stream(&v, data_ptr + s, sizeof(T), rrx_host::current_stream()));
            return v;
        }

        // device-side block gather; singleton dimensions are broadcast (reference: Array.h:579-622)
        Array_gpu<T,N> subset(const std::array<std::array<int,2>,N>& ranges) const
        {
            std::array<int,N> sd;
            for (int i=0; i<N; ++i) sd[i] = ranges[i][1] - ranges[i][0] + 1;
            Array_gpu<T,N> out(sd);
            int sub_dims[N], starts[N], spread[N]; long long st[N];
            for (int i=0; i<N; ++i)
            {
                sub_dims[i] = sd[i]; st[i] = strides[i]; spread[i] = (dims[i] == 1);
                starts[i] = ranges[i][0] - offsets[i] - 1;
            }
            if (out.ncells > 0)
                rrx_host::check(rrx_subset_nd(out.data_ptr, data_ptr, int(sizeof(T)), N, sub_dims, st, starts, spread, rrx_host::current_stream()));
            return out;
        }

        void dump(const std::string& name) const { Array<T,N> h(*this); h.dump(name); }

    private:
        // stream-ordered: the block is usable by work enqueued on the calling thread's stream from here on, and goes back to
        // the pool once the work enqueued on that stream before the release has run (no device-wide synchronisation)
        void allocate()
        {
            alloc_stream = rrx_host::current_stream();
            if (ncells > 0) rrx_host::check(rrx_malloc_async(reinterpret_cast<void**>(&data_ptr), size_t(ncells)*sizeof(T), alloc_stream));
        }
        void release()
        {
            // (the thread's stream may have been switched since the allocation, rrx_host::set_stream: ordered after both)
            if (data_ptr != nullptr && owns) rrx_free_async_ordered(data_ptr, alloc_stream, rrx_host::current_stream());
            data_ptr = nullptr;
        }
        std::array<int,N> dims;
        int ncells;
        T* data_ptr;
        std::array<int,N> strides;
        std::array<int,N> offsets;
        bool owns = true;
        void* alloc_stream = nullptr;
        template<typename, int> friend class Array;
};

template<typename T, int N>
Array<T,N>::Array(const Array_gpu<T,N>& a) : dims(a.dims), ncells(a.ncells), data(a.ncells), strides(a.strides), offsets(a.offsets)
{
    if (ncells > 0)
    {
        rrx_host::check(rrx_synchronize(rrx_host::current_stream()));
        rrx_host::check(rrx_memcpy_d2h_stream(data.data(), a.ptr(), size_t(ncells)*sizeof(T), rrx_host::current_stream()));
    }
}

// bool arrays cannot use std::vector<bool>; the reference uses Bool = signed char for that reason too.
#endif
