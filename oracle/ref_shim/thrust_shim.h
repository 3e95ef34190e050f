// oracle/ref_shim/thrust_shim.h -- the handful of Thrust algorithms and fancy iterators that the
// reference's s_filtergrid.cu uses, as serial C++14 on std::vector, so that extrema_filter_grid
// itself (the reference's own source file) runs on the CPU and pins oracle/sift_oracle.c's
// grid_filter().  Semantics follow the Thrust documentation: sort_by_key is implemented as a STABLE
// sort (thrust's radix sort for int keys is stable; its merge sort for custom comparators is stable
// too), reduce_by_key reduces runs of consecutive equal keys.
// TEST INFRASTRUCTURE ONLY.
#pragma once
#include <algorithm>
#include <cstddef>
#include <iterator>
#include <numeric>
#include <tuple>
#include <utility>
#include <vector>

namespace thrust {

using std::get;
using std::make_tuple;
using std::tuple;

template <class T>
class host_vector : public std::vector<T> {
public:
    using std::vector<T>::vector;
};
template <class T> using device_vector = host_vector<T>;
template <class T> using device_ptr = T*;
template <class T> T* device_pointer_cast(T* p) { return p; }

namespace cuda {
struct par_t { par_t on(cudaStream_t) const { return *this; } };
static const par_t par{};
}

// ---- fancy iterators ---------------------------------------------------------------------------
template <class T>
struct counting_iterator {
    using value_type = T; using reference = T; using pointer = const T*;
    using difference_type = std::ptrdiff_t; using iterator_category = std::random_access_iterator_tag;
    T v;
    T operator*() const { return v; }
    counting_iterator operator+(difference_type n) const { return {T(v + n)}; }
    difference_type operator-(const counting_iterator& o) const { return v - o.v; }
    counting_iterator& operator++() { ++v; return *this; }
    bool operator==(const counting_iterator& o) const { return v == o.v; }
    bool operator!=(const counting_iterator& o) const { return v != o.v; }
};
template <class T> counting_iterator<T> make_counting_iterator(T v) { return {v}; }

template <class T>
struct constant_iterator {
    using value_type = T; using reference = T; using pointer = const T*;
    using difference_type = std::ptrdiff_t; using iterator_category = std::random_access_iterator_tag;
    T v; difference_type pos;
    T operator*() const { return v; }
    constant_iterator operator+(difference_type n) const { return {v, pos + n}; }
    difference_type operator-(const constant_iterator& o) const { return pos - o.pos; }
    constant_iterator& operator++() { ++pos; return *this; }
};
template <class T> constant_iterator<T> make_constant_iterator(T v) { return {v, 0}; }

struct discard_sink { template <class U> const discard_sink& operator=(const U&) const { return *this; } };
struct discard_iterator {
    using value_type = int; using reference = discard_sink; using pointer = void;
    using difference_type = std::ptrdiff_t; using iterator_category = std::random_access_iterator_tag;
    discard_sink operator*() const { return {}; }
    discard_iterator operator+(difference_type) const { return {}; }
    discard_iterator& operator++() { return *this; }
};
inline discard_iterator make_discard_iterator() { return {}; }

template <class... Its>
struct zip_iterator_impl {
    using value_type = std::tuple<typename std::iterator_traits<Its>::value_type...>;
    using reference = std::tuple<typename std::iterator_traits<Its>::reference...>;
    using pointer = void;
    using difference_type = std::ptrdiff_t; using iterator_category = std::random_access_iterator_tag;
    std::tuple<Its...> its;
    template <std::size_t... I> reference deref(std::index_sequence<I...>) const { return reference(*std::get<I>(its)...); }
    template <std::size_t... I> zip_iterator_impl add(difference_type n, std::index_sequence<I...>) const
    { return {std::tuple<Its...>((std::get<I>(its) + n)...)}; }
    reference operator*() const { return deref(std::index_sequence_for<Its...>{}); }
    zip_iterator_impl operator+(difference_type n) const { return add(n, std::index_sequence_for<Its...>{}); }
    difference_type operator-(const zip_iterator_impl& o) const { return std::get<0>(its) - std::get<0>(o.its); }
    zip_iterator_impl& operator++() { *this = *this + 1; return *this; }
    bool operator==(const zip_iterator_impl& o) const { return std::get<0>(its) == std::get<0>(o.its); }
    bool operator!=(const zip_iterator_impl& o) const { return !(*this == o); }
};
template <class... Its>
zip_iterator_impl<Its...> make_zip_iterator(std::tuple<Its...> t) { return {t}; }

// ---- functors ------------------------------------------------------------------------------------
template <class T> struct multiplies { T operator()(const T& a, const T& b) const { return a * b; } };
template <class T> struct plus       { T operator()(const T& a, const T& b) const { return a + b; } };
template <class T> struct minimum    { T operator()(const T& a, const T& b) const { return b < a ? b : a; } };
template <class T> struct identity   { const T& operator()(const T& a) const { return a; } };
template <class T> struct less       { bool operator()(const T& a, const T& b) const { return a < b; } };

// ---- algorithms ----------------------------------------------------------------------------------
template <class P, class It, class T>
void fill(const P&, It b, It e, const T& v) { for (std::ptrdiff_t i = 0, n = e - b; i < n; i++) *(b + i) = v; }
template <class It>
void sequence(It b, It e) { for (std::ptrdiff_t i = 0, n = e - b; i < n; i++) *(b + i) = (int)i; }
template <class It>
void sequence(const cuda::par_t&, It b, It e) { sequence(b, e); }
template <class It, class T>
void sequence(It b, It e, T init, T step) { for (std::ptrdiff_t i = 0, n = e - b; i < n; i++) *(b + i) = init + (T)i * step; }

template <class In, class Out, class F>
Out transform(In b, In e, Out out, F f)
{
    const std::ptrdiff_t n = e - b;
    for (std::ptrdiff_t i = 0; i < n; i++) {
        typename std::iterator_traits<In>::value_type v = *(b + i);
        *(out + i) = f(v);
    }
    return out + n;
}
template <class In1, class In2, class Out, class F>
Out transform(In1 b, In1 e, In2 b2, Out out, F f)
{
    const std::ptrdiff_t n = e - b;
    for (std::ptrdiff_t i = 0; i < n; i++) *(out + i) = f(*(b + i), *(b2 + i));
    return out + n;
}

template <class KIt, class VIt, class Comp>
void sort_by_key(KIt kb, KIt ke, VIt vb, Comp comp)
{
    using K = typename std::iterator_traits<KIt>::value_type;
    using V = typename std::iterator_traits<VIt>::value_type;
    const std::ptrdiff_t n = ke - kb;
    std::vector<K> keys; std::vector<V> vals; std::vector<std::ptrdiff_t> perm(n);
    keys.reserve(n); vals.reserve(n);
    for (std::ptrdiff_t i = 0; i < n; i++) { keys.push_back(K(*(kb + i))); vals.push_back(V(*(vb + i))); perm[i] = i; }
    std::stable_sort(perm.begin(), perm.end(),
                     [&](std::ptrdiff_t a, std::ptrdiff_t b) { return comp(keys[a], keys[b]); });
    for (std::ptrdiff_t i = 0; i < n; i++) { *(kb + i) = keys[perm[i]]; *(vb + i) = vals[perm[i]]; }
}
template <class KIt, class VIt>
void sort_by_key(KIt kb, KIt ke, VIt vb)
{
    sort_by_key(kb, ke, vb, less<typename std::iterator_traits<KIt>::value_type>());
}

template <class KIt, class VIn, class KOut, class VOut>
void reduce_by_key(KIt kb, KIt ke, VIn vin, KOut kout, VOut vout)
{
    const std::ptrdiff_t n = ke - kb;
    std::ptrdiff_t r = 0;
    for (std::ptrdiff_t i = 0; i < n;) {
        auto key = *(kb + i);
        auto acc = *(vin + i);
        std::ptrdiff_t j = i + 1;
        for (; j < n && *(kb + j) == key; j++) acc = acc + *(vin + j);
        *(kout + r) = key;
        *(vout + r) = acc;
        r++; i = j;
    }
}

template <class In, class Out>
void exclusive_scan(In b, In e, Out out)
{
    typename std::iterator_traits<In>::value_type acc = 0;
    for (std::ptrdiff_t i = 0, n = e - b; i < n; i++) { auto v = *(b + i); *(out + i) = acc; acc = acc + v; }
}
template <class In, class Out>
void inclusive_scan(In b, In e, Out out)
{
    typename std::iterator_traits<In>::value_type acc = 0;
    for (std::ptrdiff_t i = 0, n = e - b; i < n; i++) { acc = acc + *(b + i); *(out + i) = acc; }
}
template <class In, class Pred>
int count_if(In b, In e, Pred p) { int c = 0; for (std::ptrdiff_t i = 0, n = e - b; i < n; i++) c += p(*(b + i)) ? 1 : 0; return c; }
template <class In>
typename std::iterator_traits<In>::value_type reduce(In b, In e)
{
    typename std::iterator_traits<In>::value_type acc = 0;
    for (std::ptrdiff_t i = 0, n = e - b; i < n; i++) acc = acc + *(b + i);
    return acc;
}
template <class In, class F>
void for_each(In b, In e, F f)
{
    for (std::ptrdiff_t i = 0, n = e - b; i < n; i++) { typename std::iterator_traits<In>::value_type v = *(b + i); f(v); }
}
template <class In, class St, class Out, class Pred>
Out copy_if(In b, In e, St stencil, Out out, Pred p)
{
    for (std::ptrdiff_t i = 0, n = e - b; i < n; i++) if (p(*(stencil + i))) { *out = *(b + i); ++out; }
    return out;
}

} // namespace thrust
