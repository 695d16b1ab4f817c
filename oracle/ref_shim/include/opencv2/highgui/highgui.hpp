// stand-in: the reference's Slic.h includes highgui for debug drawing that is compiled out (SHOW_DEBUG_VISUALISATION undefined)
#pragma once
#include <opencv2/imgproc/imgproc.hpp>
