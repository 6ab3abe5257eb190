// xavier_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
// Restates WeightServer::xavierInitializer (weight-server/weightserver.cpp:567-585):
// std::default_random_engine seeded 8888, uniform_real_distribution<float>(-1,1),
// scaled by sqrt(6 / (dim1 + dim2)).  The same libstdc++ facilities are used so
// the stream is bit-identical on this toolchain.  PARITY UNPINNED against the
// reference binary (weightserver.cpp needs boost/zmq stand-ins to build here).
#include <cmath>
#include <random>
extern "C" void orc_xavier_init(unsigned dim1, unsigned dim2, float *w) {
    std::default_random_engine dre(8888);
    std::uniform_real_distribution<float> dist(-1, 1);
    const unsigned n = dim1 * dim2;
    for (unsigned i = 0; i < n; ++i) w[i] = dist(dre);
    const float nf = std::sqrt(6.0 / (float(dim1 + dim2)));
    for (unsigned i = 0; i < n; ++i) w[i] *= nf;
}
