// dcp_errors.h -- the integer error codes of the Deciphon C API, for the library's own sources.
//
// The enum itself lives in the public header (include/deciphon_errors.h, pulled in by
// include/deciphon.h exactly where c-core/deciphon.h:34-116 has it).
#pragma once

#include "../../include/deciphon_errors.h"

#ifdef __cplusplus
extern "C"
#endif
char const *dcp_error_string(int error_code); // c-core/error.c:94-101
