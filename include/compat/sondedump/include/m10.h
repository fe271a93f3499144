/* compat shim: the reference includes "sondedump/include/m10.h" (src/decode/decoder.hpp:6-14) */
#include "../../../sonde_abi.h"
