// grid.hpp — row-major 2-D output container, the destination of the render call.
// Same role and memory layout as the reference's cutrace::grid<T> (inc/grid.hpp:20-332:
// contiguous T[w*h], row i at data(i) = base + i*w); only what the render path and the
// image writers need.
#ifndef CUTRACE_AMD_GRID_HPP
#define CUTRACE_AMD_GRID_HPP
#include <cstddef>
#include <vector>

namespace cutrace {
struct vector { float x, y, z; };  // inc/vector.hpp:25-28

template <typename T>
class grid {
public:
  grid() = default;
  grid(size_t w, size_t h) { resize(w, h); }
  void resize(size_t w, size_t h) { w_ = w; h_ = h; ext_ = nullptr; buf_.assign(w * h, T{}); }  // inc/grid.hpp:276-278
  // storage provided by the caller (e.g. a part of a page-locked ctr_frame_alloc block): no copy, no fill
  void adopt(T *p, size_t w, size_t h) { w_ = w; h_ = h; buf_.clear(); ext_ = p; }
  size_t cols() const { return w_; }
  size_t rows() const { return h_; }
  size_t elems() const { return w_ * h_; }
  T *data(size_t row = 0) { return base() + row * w_; }                            // inc/grid.hpp:267
  const T *data(size_t row = 0) const { return base() + row * w_; }
  T &raw(size_t i) { return base()[i]; }                                           // inc/grid.hpp:309
  const T &raw(size_t i) const { return base()[i]; }
  T &at(size_t x, size_t y) { return base()[y * w_ + x]; }
  const T &at(size_t x, size_t y) const { return base()[y * w_ + x]; }

private:
  T *base() { return ext_ ? ext_ : buf_.data(); }
  const T *base() const { return ext_ ? ext_ : buf_.data(); }
  size_t w_ = 0, h_ = 0;
  std::vector<T> buf_;
  T *ext_ = nullptr;
};
}  // namespace cutrace
#endif
