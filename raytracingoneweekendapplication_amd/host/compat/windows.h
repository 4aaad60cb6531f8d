// compat/windows.h -- lets a reference-style main.cpp that still carries its Win32 shell (main.cpp:22,32-37,
// 473-483: MultiByteToWideChar, ShellExecuteW, MessageBox) compile on Linux against the drop-in headers.
// The calls are no-ops that report failure/success the way the program expects; nothing here is on the
// rendering path.  Also provides the unqualified max()/min() the reference relies on from <windows.h>
// (Camera.txt:249).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>

using std::max;
using std::min;

typedef void* HINSTANCE;
typedef void* HWND;
typedef intptr_t INT_PTR;
typedef unsigned int UINT;
typedef const wchar_t* LPCWSTR;
#define CP_UTF8 65001
#define SW_SHOWNORMAL 1
#define MB_OK 0x0u
#define MB_ICONERROR 0x10u

inline int MultiByteToWideChar(UINT, unsigned long, const char* src, int, wchar_t* dst, int dst_len) {
    const int n = int(std::strlen(src)) + 1;
    if (dst)
        for (int i = 0; i < n && i < dst_len; i++) dst[i] = wchar_t(static_cast<unsigned char>(src[i]));
    return n;
}
inline HINSTANCE ShellExecuteW(HWND, LPCWSTR, LPCWSTR, LPCWSTR, LPCWSTR, int) { return reinterpret_cast<HINSTANCE>(intptr_t(33)); }  // "> 32" = success
inline int MessageBox(HWND, LPCWSTR, LPCWSTR, UINT) { return 0; }
