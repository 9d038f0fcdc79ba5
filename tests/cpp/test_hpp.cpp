// Host-side C++ mirror (include/cssm_pf.hpp) driven exactly like the reference's examples:
// Filter(mod, resample).llFilter(data, n) -- examples/Filtering.scala:23-32.  Prints the log-likelihood
// in hex so that the Python test can compare it bit for bit with the ctypes binding and the oracle.
#include <cstdio>
#include <vector>
#include "cssm_pf.hpp"

int main(int argc, char** argv) {
  using namespace cssm;
  const uint64_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 4096;
  try {
    // C2: Model.poisson(Sde.ouProcess(1)) |+| Model.seasonal(24, 1, Sde.ouProcess(2))
    UnparamModel um = Model::poisson(Sde::ouProcess(1)) | Model::seasonal(24, 1, Sde::ouProcess(2));
    Parameters p = Parameters{{std::nullopt, SdeParameter::ouParameter({0.0}, {1.0}, {0.2}, {0.0}, {0.3})}} |
                   Parameters{{std::nullopt, SdeParameter::ouParameter({0.0}, {1.0}, {0.2}, {1.0}, {0.3})}};
    ParamModel mod(um, p);
    std::vector<Data> data;
    const double ys[] = {2, 1, 4, 0, 3, 2, 5, 1};
    for (int i = 0; i < 8; ++i) data.push_back({(double)i, i == 3 ? std::optional<double>{} : std::optional<double>{ys[i]}});
    Filter f(mod, n, 20260101);
    const double ll = f.llFilter(data);
    auto [ll2, path] = f.filter(data);
    PfState s = f.initialiseState(0.0);
    for (auto& d : data) s = f.stepFilter(s, d);
    std::printf("ll %a\nll_filter %a\nll_stream %a\npath_len %zu\ness %d\n", ll, ll2, s.ll, path.size(), s.ess);
    // the Resample[A] seam (package.scala:23) on host vectors: systematic with a given u, stratified / multinomial by (seed, step)
    {
      Vec w; std::vector<int> ids;
      for (int i = 0; i < 1000; ++i) { w.push_back((i * 37 % 101) / 101.0 + (i == 500 ? 5.0 : 0.0)); ids.push_back(i); }
      auto a = Resampling::systematicResampling(ids, w, 0.25);
      auto b = Resampling::stratifiedResampling(ids, w, 20260101, 3);
      auto c = Resampling::multinomialResampling(ids, w, 20260101, 3);
      unsigned long long ha = 0, hb = 0, hc = 0;
      for (size_t i = 0; i < a.size(); ++i) { ha = ha * 1000003ull + a[i]; hb = hb * 1000003ull + b[i]; hc = hc * 1000003ull + c[i]; }
      std::printf("resample_len %zu\nresample_sys %llu\nresample_strat %llu\nresample_multi %llu\n", a.size(), ha, hb, hc);
    }
    // error path: a Gaussian model without its scale parameter must throw (Model.scala:250)
    try {
      ParamModel bad(Model::linear(Sde::brownianMotion(1)), Parameters{{std::nullopt, SdeParameter::brownianParameter({0.0}, {1.0}, {1.0})}});
      Filter g(bad, 16);
      std::printf("error_not_raised\n");
    } catch (const Error& e) { std::printf("error_code %d\n", e.code); }
  } catch (const std::exception& e) {
    std::printf("exception %s\n", e.what());
    return 1;
  }
  return 0;
}
