/* Version of the cudecomp.h API this library implements (mirrors cuDecomp 0.7.0,
 * reference include/cudecomp_version.h:20-26) and the layout versions of its three POD structs. */
#ifndef CUDECOMP_VERSION_H
#define CUDECOMP_VERSION_H

#define CUDECOMP_MAJOR 0
#define CUDECOMP_MINOR 7
#define CUDECOMP_PATCH 0

#define CUDECOMP_GRID_DESC_CONFIG_VERSION 1
#define CUDECOMP_GRID_DESC_AUTOTUNE_OPTIONS_VERSION 1
#define CUDECOMP_PENCIL_INFO_VERSION 1

#endif
