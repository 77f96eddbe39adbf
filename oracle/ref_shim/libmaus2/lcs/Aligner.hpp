// ORACLE SUPPORT (test infrastructure): forwards to the libmaus2 stand-in, see libmaus2/shim.hpp
#include <libmaus2/shim.hpp>
