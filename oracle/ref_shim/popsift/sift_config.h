#include "../sift_config.h"
