/* Float / Bool of the build, same switches as the reference (/root/reference/include/types.h:13-25):
 * -DRTE_USE_SP selects float; Bool is signed char (RTE_USE_CBOOL, the only setting the reference's configs use). */
#ifndef TYPES_H
#define TYPES_H
#include <cfloat>

using Bool = signed char;

#ifdef RTE_USE_SP
using Float = float;
const Float Float_epsilon = FLT_EPSILON;
#define RRX_SFX(name) name##_f32
#else
using Float = double;
const Float Float_epsilon = DBL_EPSILON;
#define RRX_SFX(name) name##_f64
#endif

using Int = unsigned long long;
#endif
