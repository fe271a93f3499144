/* compat shim: the reference includes "sondedump/include/imet4.h" (src/decode/decoder.hpp:6-14) */
#include "../../../sonde_abi.h"
