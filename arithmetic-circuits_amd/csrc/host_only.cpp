// host_only.cpp -- the pure-host entry points of include/acx.h (acx_strerror, acx_last_error, acx_version, acx_circuit_*)
// in a library with NO HIP in it: the same source text as libacx.so's (abi_common.inc.h, circuit_abi.inc.h, circuit_host.h,
// host_field.h), compiled by g++ with -fsanitize=address,undefined for tests/test_host_sanitized.py.  The marshalled gate
// list is an untrusted token stream (the reference's equivalent failure is `panic`,
// /root/reference/src/Circuit/Arithmetic.hs:128,137); this build is where out-of-bounds reads, overflows and leaks in the
// code that parses it would show.  Test infrastructure: nothing ships from here.
#include "abi_common.inc.h"

extern "C" {
#include "circuit_abi.inc.h"
}
