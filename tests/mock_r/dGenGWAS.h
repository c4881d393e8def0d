// Syntax-check mock for r_shim/gpu_shim.cpp (test infrastructure; tests/test_cpu_host.py::test_r_shim_compiles_against_mock).
//
// R, gdsfmt and the SNPRelate sources are not in this repository's build image, so the shim cannot be compiled against
// them here (r_shim/check_syntax.sh does that where they exist).  This header only DECLARES, in this file's own words, the
// handful of names the shim uses from <Rinternals.h>, gdsfmt's R_GDS_CPP.h and SNPRelate's dGenGWAS.h -- same names, same
// parameter types (reference: src/dGenGWAS.h:94-108 CdBaseWorkSpace, :272-281 CProgress, :295-317 CGenoReadBySNP, :428
// TimeToStr, :796 MCWorkingGeno, :841-847 helpers) -- so that `g++ -fsyntax-only` catches typos and type errors in the
// never-compiled file.  Nothing here is linked, run, or used to build any part of the reference.
#pragma once
#include <cstddef>
#include <cstdint>
#include <exception>
#include <vector>

// ---- <Rinternals.h> -----------------------------------------------------------------------------------------------
typedef struct SEXPREC *SEXP;
typedef ptrdiff_t R_xlen_t;
enum { INTSXP = 13, REALSXP = 14, VECSXP = 19 };
#ifndef TRUE
#define TRUE 1
#endif
extern SEXP R_NilValue;
extern double R_NaN;
extern int R_NaInt;
#define NA_INTEGER R_NaInt
SEXP Rf_protect(SEXP);
void Rf_unprotect(int);
#define PROTECT(s) Rf_protect(s)
#define UNPROTECT(n) Rf_unprotect(n)
SEXP Rf_allocVector(unsigned int, R_xlen_t);
SEXP Rf_allocMatrix(unsigned int, int, int);
SEXP Rf_ScalarReal(double);
SEXP Rf_GetOption1(SEXP);
SEXP Rf_install(const char *);
SEXP Rf_coerceVector(SEXP, unsigned int);
int Rf_asInteger(SEXP);
int Rf_asLogical(SEXP);
int Rf_length(SEXP);
R_xlen_t Rf_xlength(SEXP);
int Rf_isNull(SEXP);
void Rf_error(const char *, ...);
void Rprintf(const char *, ...);
void R_CheckUserInterrupt(void);
double *REAL(SEXP);
int *INTEGER(SEXP);
const char *CHAR(SEXP);
SEXP STRING_ELT(SEXP, R_xlen_t);
SEXP VECTOR_ELT(SEXP, R_xlen_t);
SEXP SET_VECTOR_ELT(SEXP, R_xlen_t, SEXP);

// ---- gdsfmt: R_GDS_CPP.h / dType.h ----------------------------------------------------------------------------------
typedef uint8_t C_UInt8;
typedef int32_t C_Int32;
typedef int64_t C_Int64;
typedef void *PdGDSObj;
typedef void *PdGDSFolder;
typedef void *PdAbstractArray;
enum C_SVType { svFloat64 = 13 };
#define COREARRAY_DLL_EXPORT
#define COREARRAY_DLL_LOCAL
PdGDSFolder GDS_R_SEXP2FileRoot(SEXP);
PdGDSObj GDS_Node_Path(PdGDSFolder, const char *, int);
void GDS_Array_AppendData(PdAbstractArray, ptrdiff_t, const void *, enum C_SVType);
void GDS_SetError(const char *);
namespace CoreArray {
class ErrCoreArray : public std::exception {
public:
    ErrCoreArray(const char *fmt, ...);
    const char *what() const noexcept override;
};
}  // namespace CoreArray
using CoreArray::ErrCoreArray;
#define COREARRAY_TRY \
    SEXP rv_ans = R_NilValue; \
    bool has_error = false; \
    try {
#define COREARRAY_CATCH \
    } catch (std::exception &E) { GDS_SetError(E.what()); has_error = true; } \
    catch (const char *E) { GDS_SetError(E); has_error = true; } \
    catch (...) { GDS_SetError("unknown error!"); has_error = true; } \
    if (has_error) Rf_error("%s", "error"); \
    return rv_ans;

// ---- SNPRelate: dGenGWAS.h --------------------------------------------------------------------------------------------
namespace GWAS {
enum TTypeGenoDim { RDim_Sample_X_SNP = 0, RDim_SNP_X_Sample = 1 };
class CdBaseWorkSpace {
public:
    C_Int32 SampleNum() const;
    C_Int32 SNPNum() const;
};
class CProgress {
public:
    CProgress();
    CProgress(C_Int64 count);
    void Forward(C_Int64 val);
};
class CGenoReadBySNP {
public:
    CGenoReadBySNP(int num_thread, CdBaseWorkSpace &space, size_t max_cnt_snp, C_Int64 progress_count, bool mem_load,
                   TTypeGenoDim dim = RDim_Sample_X_SNP);
    ~CGenoReadBySNP();
    void Init();
    bool Read(C_UInt8 *OutGeno);
    void ProgressForward(C_Int64 val);
    size_t Count() const;
};
class CMultiCoreWorkingGeno {
public:
    CdBaseWorkSpace &Space();
};
extern CMultiCoreWorkingGeno MCWorkingGeno;
const char *TimeToStr();
SEXP RGetListElement(SEXP list, const char *name);
bool SEXP_Verbose(SEXP Verbose);
void CachingSNPData(const char *Msg, bool Verbose);
}  // namespace GWAS
