#include "cusim.h"
