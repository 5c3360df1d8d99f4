# Sourced by the timing-experiment scripts (GPU box only).  An instrumented or re-flagged build REPLACES objects of the production libv2x_amd.so (same ABI
# number, garbage results): whatever happens -- normal end, an error, Ctrl-C, a timeout's SIGTERM -- the EXIT trap removes those objects and rebuilds the
# product (an interrupted run used to leave a wrong-results library in place).
export V2X_ALLOW_PROBE_BUILD=1     # the loader refuses a library with an instrumented object (negative v2x_abi_version) unless asked
BASE_FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
PROBE_TOUCHED=""
probe_restore() {
    for n in $PROBE_TOUCHED; do rm -f v2x-sim_amd/csrc/build/$n.o v2x-sim_amd/csrc/build/${n}_probe.hip; done
    make -s -C v2x-sim_amd/csrc > /dev/null 2>&1
}
trap probe_restore EXIT
trap 'exit 130' INT TERM HUP
# probe_build <name> "<extra flags>": object <name>.o from the GENERATED instrumented copy (tools/probes/gen_probe.sh)
probe_build() {
    PROBE_TOUCHED="$PROBE_TOUCHED $1"
    rm -f v2x-sim_amd/csrc/build/$1.o
    make -s -C v2x-sim_amd/csrc PROBE=$1 FLAGS="$BASE_FLAGS $2" > /dev/null 2>&1
}
# prod_build <name> "<extra flags>": the PRODUCTION source of <name> with extra -D flags
prod_build() {
    PROBE_TOUCHED="$PROBE_TOUCHED $1"
    rm -f v2x-sim_amd/csrc/build/$1.o
    make -s -C v2x-sim_amd/csrc FLAGS="$BASE_FLAGS $2" > /dev/null 2>&1
}
