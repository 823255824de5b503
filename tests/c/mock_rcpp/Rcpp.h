// tests/c/mock_rcpp/Rcpp.h -- TEST INFRASTRUCTURE.  A mock of exactly the Rcpp / R API declarations bindings/r/CogapsHip.cpp uses, so that
// the glue can meet a compiler front end (g++ -fsyntax-only) in an image without R.  It declares shapes, not behaviour: nothing here is
// ever linked or run, and it pins nothing about R or Rcpp themselves.  Written for this repository (no Rcpp text).
#pragma once
#include <string>
#include <type_traits>
#include <vector>
#include <stdint.h>
struct SEXPREC;
typedef SEXPREC *SEXP;
#define RcppExport extern "C"
#define BEGIN_RCPP try {
#define END_RCPP } catch (...) { } return (SEXP)0;
extern "C" int Rf_isNull(SEXP);
namespace Rcpp {
// what `list["name"]`, `list[i]` and `s4.slot("name")` give: reads convert to anything, writes take anything
struct Named;
struct Proxy {
    template <class T, class = typename std::enable_if<!std::is_same<T, Named>::value>::type> operator T() const;
    template <class T> Proxy &operator=(const T &);
};
struct RObject { RObject(); RObject(SEXP); operator SEXP() const; RObject &operator=(SEXP); };
struct RNGScope { RNGScope(); ~RNGScope(); };
template <class T> struct Nullable { Nullable(); Nullable(SEXP); bool isNotNull() const; bool isNull() const; };
struct NamedValue { };
struct Named { explicit Named(const char *); template <class T> NamedValue operator=(const T &) const; };
struct NumericMatrix {
    NumericMatrix(); NumericMatrix(int nrow, int ncol); NumericMatrix(const Nullable<NumericMatrix> &);
    int nrow() const; int ncol() const;
    double &operator()(int i, int j); double operator()(int i, int j) const;
};
struct NumericVector { NumericVector(); template <class A, class B> static NumericVector create(const A &, const B &); };
struct CharacterVector { CharacterVector(); CharacterVector(const Nullable<CharacterVector> &); void push_back(const std::string &); };
struct List {
    List(); explicit List(unsigned n);
    Proxy operator[](const char *) const; Proxy operator[](unsigned) const; Proxy operator[](int) const;
    template <class... A> static List create(const A &...);
};
struct S4 { S4(); S4(const Proxy &); Proxy slot(const char *) const; };
template <class T, class U> T as(const U &);
template <class T> SEXP wrap(const T &);
void stop(const std::string &);
void checkUserInterrupt();
}
