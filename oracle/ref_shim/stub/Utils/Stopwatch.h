// Stand-in for Core/Utils/Stopwatch.h (timing macros that talk to a UDP logger), TEST INFRASTRUCTURE ONLY.
#pragma once
#define TICK(name) ((void)0)
#define TOCK(name) ((void)0)
