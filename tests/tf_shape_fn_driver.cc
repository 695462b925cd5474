// Test driver (tests/tf_api_stub/README.md): runs the shape functions the TF glue registers, on cases read from
// stdin, through the behavioural InferenceContext stand-in. One case per line:
//   <op> <transform_type|-> <source shape> <points shape> <grid_shape input>
// shapes: "?" (unknown rank) or comma-separated dims with "?" for an unknown one, "-" for rank 0;
// grid_shape input: "none" (op has no such input), "v:<dims>" constant value, "r:<n>" unknown values of known
// length n, "r:?" unknown length. Prints "OK <shape>" or "ERR <message>" per case.
#include <iostream>
#include <sstream>

#include "../tensorflow-nufft_amd/csrc/tf_glue/nufft_tf_ops.cc"

using tensorflow::shape_inference::InferenceContext;
using tensorflow::shape_inference::ShapeHandle;

static std::vector<int64_t> parse_dims(const std::string& s) {
  std::vector<int64_t> d;
  if (s == "-") return d;
  std::stringstream ss(s);
  std::string tok;
  while (std::getline(ss, tok, ',')) d.push_back(tok == "?" ? -1 : std::stoll(tok));
  return d;
}
static ShapeHandle parse_shape(const std::string& s) {
  if (s == "?") return {};
  return {true, parse_dims(s)};
}

int main() {
  std::string line;
  while (std::getline(std::cin, line)) {
    if (line.empty()) continue;
    std::stringstream ss(line);
    std::string op, ttype, src, pts, grid;
    ss >> op >> ttype >> src >> pts >> grid;
    const auto it = tensorflow::StubOpRegistry().find(op);
    if (it == tensorflow::StubOpRegistry().end()) { std::cout << "ERR no such op " << op << "\n"; continue; }
    InferenceContext c;
    c.inputs = {parse_shape(src), parse_shape(pts)};
    if (ttype != "-") c.string_attrs["transform_type"] = ttype;
    if (grid != "none") {
      if (grid.rfind("v:", 0) == 0) {
        const std::vector<int64_t> v = parse_dims(grid.substr(2));
        c.inputs.push_back({true, {(int64_t)v.size()}});
        c.input_values[2] = v;
      } else if (grid == "r:?") {
        c.inputs.push_back({true, {-1}});
      } else {
        c.inputs.push_back({true, {std::stoll(grid.substr(2))}});
      }
    }
    const tensorflow::Status s = it->second(&c);
    if (s.ok()) std::cout << "OK " << c.DebugString(c.outputs[0]) << "\n";
    else std::cout << "ERR " << s.message() << "\n";
  }
  return 0;
}
