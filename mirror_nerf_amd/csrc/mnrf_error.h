// mnrf_error.h -- error reporting shared by the C-ABI translation units.
// The ABI never throws: functions return a negative code and keep a thread-local message
// (include/mnrf.h, mnrf_last_error).  Defined once in mnrf_render.hip.
#pragma once
int mnrf_fail(int code, const char* msg);
int mnrf_check_launch(const char* where);
