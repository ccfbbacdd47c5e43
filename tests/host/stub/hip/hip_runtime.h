// Stand-in for <hip/hip_runtime.h> when the device routines are compiled for the host by
// tests/host/contours_host.cpp (which supplies the handful of names they use).
#pragma once
