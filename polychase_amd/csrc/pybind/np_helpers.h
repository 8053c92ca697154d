// np_helpers.h -- numpy <-> plain C++ containers (the reference uses pybind11/eigen.h, cvnp and
// cpp/pybind11_extension.h; Eigen and OpenCV are not dependencies here).
#pragma once

#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <array>
#include <cstring>
#include <vector>

namespace py = pybind11;

using F32Array = py::array_t<float, py::array::c_style | py::array::forcecast>;
using U32Array = py::array_t<uint32_t, py::array::c_style | py::array::forcecast>;

// std::vector<std::array<float, D>>  <->  (N, D) float32   (pybind11_extension.h:24-56)
template <size_t D>
inline py::array_t<float> VecToNumpy(const std::vector<std::array<float, D>>& v) {
    py::array_t<float> a({static_cast<py::ssize_t>(v.size()), static_cast<py::ssize_t>(D)});
    if (!v.empty()) std::memcpy(a.mutable_data(), v.data(), v.size() * D * sizeof(float));
    return a;
}
template <size_t D>
inline std::vector<std::array<float, D>> NumpyToVec(const F32Array& a) {
    if (a.ndim() != 2 || a.shape(1) != static_cast<py::ssize_t>(D))
        throw py::value_error("expected an array of shape (N, " + std::to_string(D) + ")");
    std::vector<std::array<float, D>> v(static_cast<size_t>(a.shape(0)));
    if (!v.empty()) std::memcpy(v.data(), a.data(), v.size() * D * sizeof(float));
    return v;
}
template <typename T>
inline py::array_t<T> Vec1ToNumpy(const std::vector<T>& v) {
    py::array_t<T> a(static_cast<py::ssize_t>(v.size()));
    if (!v.empty()) std::memcpy(a.mutable_data(), v.data(), v.size() * sizeof(T));
    return a;
}
template <typename T>
inline std::vector<T> NumpyToVec1(const py::array_t<T, py::array::c_style | py::array::forcecast>& a) {
    std::vector<T> v(static_cast<size_t>(a.size()));
    if (!v.empty()) std::memcpy(v.data(), a.data(), v.size() * sizeof(T));
    return v;
}
template <size_t N>
inline py::array_t<float> ArrToNumpy(const std::array<float, N>& v) {
    py::array_t<float> a(static_cast<py::ssize_t>(N));
    std::memcpy(a.mutable_data(), v.data(), N * sizeof(float));
    return a;
}
template <size_t N>
inline std::array<float, N> NumpyToArr(const F32Array& a) {
    if (static_cast<size_t>(a.size()) != N) throw py::value_error("expected " + std::to_string(N) + " floats");
    std::array<float, N> v;
    std::memcpy(v.data(), a.data(), N * sizeof(float));
    return v;
}
