// dcp_errors.h -- the integer error codes of the Deciphon C API, for the library's own sources.
//
// The enum itself lives in the public header, include/deciphon.h, exactly where the reference has
// it (c-core/deciphon.h:34-116).
#pragma once

#include "../../include/deciphon.h"
