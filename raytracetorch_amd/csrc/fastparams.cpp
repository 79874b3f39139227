// _fastparams — one call that tells a scene whether any of its parameters changed since its surface program was packed.
//
// The reference re-reads every nn.Parameter on every surface visit (geom/primitives.py:263, geom/transform.py:25,
// elements/lens.py:28-29 are read inside scene/sequential.py:17-34); here the parameters are packed into records once
// and the pack is reused while nothing changed.  "Nothing changed" has to be decided before every launch, for scenes
// of a few hundred small tensors, and reading an attribute of each from Python costs more than the launch.  snapshot()
// walks the list in C++:
//   storage   hash of every tensor's data pointer, element count and device: `p.data = other`, `.to()`, a resized tensor
//   grad      whether any tensor requires grad (decides between the plain and the differentiable trace)
//   versions  hash of the version counters of the tensors that do not live in host memory (their values cannot be read
//             without a synchronisation; writes through the tensor bump the counter)
//   values    the raw bytes of the host-resident tensors behind one another: a write through `.data`, which leaves the
//             version counter alone (the reference's documented parameter sweep, project.py:18-23), still shows
#include <torch/csrc/autograd/python_variable.h>
#include <torch/extension.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace {

inline uint64_t mix(uint64_t h, uint64_t v) {
  h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
  return h * 0xFF51AFD7ED558CCDull;
}

py::tuple snapshot(const py::list& params) {
  uint64_t storage = 0x1234567ull, versions = 0x89ABCDEull;
  bool grad = false;
  std::string values;
  values.reserve(params.size() * 16);
  for (const py::handle& item : params) {
    if (!THPVariable_Check(item.ptr())) throw py::type_error("snapshot() takes a list of tensors");
    const at::Tensor& t = THPVariable_Unpack(item.ptr());   // borrowed: no reference count traffic per tensor
    storage = mix(storage, reinterpret_cast<uint64_t>(t.data_ptr()));
    storage = mix(storage, static_cast<uint64_t>(t.numel()) * 64u + static_cast<uint64_t>(t.device().type()) * 8u +
                               static_cast<uint64_t>(t.scalar_type()));
    grad = grad || t.requires_grad();
    if (t.device().is_cpu()) {
      if (t.is_contiguous()) {
        values.append(static_cast<const char*>(t.data_ptr()), t.nbytes());
      } else {
        const at::Tensor c = t.detach().contiguous();
        values.append(static_cast<const char*>(c.data_ptr()), c.nbytes());
      }
    } else {
      versions = mix(versions, static_cast<uint64_t>(t._version()));
    }
  }
  return py::make_tuple(storage, grad, versions, py::bytes(values));
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "parameter snapshots for the packed-program cache of raytracetorch_amd scenes";
  m.def("snapshot", &snapshot, "(storage hash, any requires_grad, version hash of device tensors, bytes of host tensors)");
}
