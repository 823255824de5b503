// tests/c/mock_rcpp/R_ext/Rdynload.h -- TEST INFRASTRUCTURE, see ../Rcpp.h: the registration types bindings/r/CogapsHip.cpp names.
#pragma once
struct SEXPREC;
typedef struct _DllInfo DllInfo;
typedef void *(*DL_FUNC)();
typedef struct { const char *name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef R_CallMethodDef R_ExternalMethodDef;
typedef struct { const char *name; DL_FUNC fun; int numArgs; void *types; } R_CMethodDef;
typedef R_CMethodDef R_FortranMethodDef;
#ifndef FALSE
#define FALSE 0
#endif
extern "C" int R_registerRoutines(DllInfo *, const R_CMethodDef *, const R_CallMethodDef *, const R_FortranMethodDef *, const R_ExternalMethodDef *);
extern "C" int R_useDynamicSymbols(DllInfo *, int);
