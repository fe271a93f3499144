/* compat shim: the reference includes "sondedump/include/c50.h" (src/decode/decoder.hpp:6-14) */
#include "../../../sonde_abi.h"
