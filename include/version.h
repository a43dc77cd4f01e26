/* version.h -- version of the MI355X build; HPRLP_VERSION_STRING tracks the reference release
 * whose boundary is mirrored (reference include/version.h:16-19). */
#ifndef HPRLP_VERSION_H
#define HPRLP_VERSION_H
#define HPRLP_VERSION_MAJOR 0
#define HPRLP_VERSION_MINOR 1
#define HPRLP_VERSION_PATCH 2
#define HPRLP_VERSION_STRING "0.1.2"
#define HPRLP_BACKEND_STRING "hip-gfx950"
#ifdef __cplusplus
extern "C" {
#endif
static inline const char *hprlp_get_version(void) { return HPRLP_VERSION_STRING; }
static inline int hprlp_get_version_major(void) { return HPRLP_VERSION_MAJOR; }
static inline int hprlp_get_version_minor(void) { return HPRLP_VERSION_MINOR; }
static inline int hprlp_get_version_patch(void) { return HPRLP_VERSION_PATCH; }
#ifdef __cplusplus
}
#endif
#endif
