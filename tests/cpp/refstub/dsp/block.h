// TEST-ONLY skeleton of SDR++'s dsp/block.h (absent here): just enough surface for /root/reference/src/decode/decoder.hpp
// to compile so that tests/test_ref_boundary.py can instantiate the reference's own radiosonde::Decoder<> against this
// repo's C ABI.  Boundary evidence, not an oracle and not a reference build; nothing in the product includes it.
#pragma once
#include <cassert>
#include <cmath>
#include <cstddef>
#include <sstream>
#include <string>
namespace dsp {
template <class T> struct stream { T *readBuf = nullptr; int pending = -1; int read() { return pending; } void flush() { pending = -1; } };
struct block { bool _block_init = false; virtual ~block() {} virtual int run() = 0; void stop() {}
	template <class S> void registerInput(S *) {} template <class S> void unregisterInput(S *) {} };
}
