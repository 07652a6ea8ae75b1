// TEST INFRASTRUCTURE: host stand-ins for the two rocPRIM primitives sort_unique.hip falls back to (lists beyond its own sort's range), with rocPRIM's
// calling convention — a null temporary-storage pointer is the size query — so that the CPU build of that file (tests/emul/build_emul.py) is complete.
#pragma once
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

namespace rocprim {
template <typename T>
struct counting_iterator {
    T base;
    explicit counting_iterator(T b) : base(b) {}
    T operator[](size_t i) const { return (T)(base + (T)i); }
};
template <typename T>
struct plus {
    T operator()(T a, T b) const { return a + b; }
};
template <typename K, typename VI, typename V>
hipError_t radix_sort_pairs(void* tmp, size_t& bytes, const K* keys_in, K* keys_out, VI values_in, V* values_out, size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t) {
    if (!tmp) {
        bytes = 16;
        return hipSuccess;
    }
    const K mask = (end_bit - begin_bit >= 8 * sizeof(K)) ? ~(K)0 : ((((K)1 << (end_bit - begin_bit)) - 1) << begin_bit);
    std::vector<size_t> p(n);
    std::iota(p.begin(), p.end(), (size_t)0);
    std::stable_sort(p.begin(), p.end(), [&](size_t a, size_t b) { return (keys_in[a] & mask) < (keys_in[b] & mask); });
    std::vector<K> k(n);
    std::vector<V> v(n);
    for (size_t i = 0; i < n; ++i) {
        k[i] = keys_in[p[i]];
        v[i] = (V)values_in[p[i]];
    }
    std::copy(k.begin(), k.end(), keys_out);
    std::copy(v.begin(), v.end(), values_out);
    return hipSuccess;
}
template <typename T, typename Op>
hipError_t inclusive_scan(void* tmp, size_t& bytes, const T* in, T* out, size_t n, Op op, hipStream_t) {
    if (!tmp) {
        bytes = 16;
        return hipSuccess;
    }
    T acc{};
    for (size_t i = 0; i < n; ++i) {
        acc = i == 0 ? in[0] : op(acc, in[i]);
        out[i] = acc;
    }
    return hipSuccess;
}
}  // namespace rocprim
