/* Go evaluates constant expressions exactly and rounds once; these are the resulting float64 values
 * (effects/effects.go:48-56, fft/fft.go:24-27, spatializer/spatializer.go:12-25). Generated, see DESIGN.md. */
#ifndef GDG_GO_CONSTS_H
#define GDG_GO_CONSTS_H
#define GO_MATH_DEGREE_TO_RADIANS 0x1.1df46a2529d39p-6 /* 0.017453292519943295 */
#define GO_MATH_PI_THOUSANDTH 0x1.9bc65b68b71c3p-9 /* 0.0031415926535897933 */
#define GO_MATH_QUARTER_PI 0x1.921fb54442d18p-1 /* 0.7853981633974483 */
#define GO_MATH_TWO_OVER_PI 0x1.45f306dc9c883p-1 /* 0.6366197723675814 */
#define GO_MATH_TWO_PI 0x1.921fb54442d18p+2 /* 6.283185307179586 */
#define GO_MATH_TWO_PI_FIFTH 0x1.41b2f769cf0e0p+0 /* 1.2566370614359172 */
#define GO_MATH_TWO_PI_HUNDREDTH 0x1.015bf9217271ap-4 /* 0.06283185307179587 */
#define GO_MATH_INV_SQRT_2 0x1.6a09e667f3bcdp-1 /* 0.7071067811865476 */
#define GO_HALF_EFFECTIVE_DISTANCE 0x1.b851eb851eb85p-4 /* 0.1075 */
#define GO_GROUP_DELAY_OVER_EFFECTIVE_DISTANCE 0x1.80124a0386436p-9 /* 0.002930232558139535 */
#endif
