// Error reporting shared by the translation units of librt_amd.so.
#pragma once

// Stores msg as the calling thread's rt_last_error() text and returns code.
int rt_fail_message(int code, const char* msg);
