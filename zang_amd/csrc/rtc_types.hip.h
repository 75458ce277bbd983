// rtc_types.hip.h -- under hiprtc there is no <stdint.h>: the fixed-width types live in __hip_internal.
#pragma once
#if defined(__HIPCC_RTC__)
typedef __hip_internal::int8_t int8_t;
typedef __hip_internal::uint8_t uint8_t;
typedef __hip_internal::int16_t int16_t;
typedef __hip_internal::uint16_t uint16_t;
typedef __hip_internal::int32_t int32_t;
typedef __hip_internal::uint32_t uint32_t;
typedef __hip_internal::int64_t int64_t;
typedef __hip_internal::uint64_t uint64_t;
typedef unsigned long uintptr_t;
#endif
#if defined(__HIPCC_RTC__)
#ifndef INT32_MIN
#define INT32_MIN (-2147483647 - 1)
#define INT32_MAX 2147483647
#define UINT32_MAX 4294967295u
#endif
#endif
