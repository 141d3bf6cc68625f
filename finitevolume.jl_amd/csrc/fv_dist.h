// Distributed (row-block) declarations live in fv_internal.h; kept as a separate include point.
#pragma once
#include "fv_internal.h"
