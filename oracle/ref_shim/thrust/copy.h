#include "../thrust_shim.h"
