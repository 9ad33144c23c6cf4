#!/bin/bash
# VALU per wave of the trace kernel for one generation variant, per library build.  Usage: sq_variant_libs.sh <variant> name=lib.so ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=$1; shift
for arm in "$@"; do
  name=${arm%%=*}; lib=${arm#*=}
  OPTRACE_AMD_LIB=$R/$lib VARIANTS=$V bash $R/tools/sq_gen_variants.sh | sed "s/^/$name: /"
done
