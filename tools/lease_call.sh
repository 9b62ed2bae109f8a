#!/bin/bash
# One GPU lease = one recipe (run through gpurun: `gpurun --timeout S -- bash tools/lease_call.sh <recipe> <tag> [args]`).  Output under gpurun_out/<tag>/;
# copy what should be judged into profiles/.  Replaces the per-call scripts of earlier rounds (tools/r05_call_[a-w].sh, git history).
#   suite   <tag> [pytest args]         the GPU test suite (-m gpu), timed
#   bench   <tag> [bench.py args]       one bench.py line (default: the driver's command)
#   ab      <tag> ENV=V[,V..] [bench.py args]   headline A/B of one IRSDE_TUNING variable, 3 timed steps per value
#   profile <tag> [bench.py args]       rocprofv3 kernel trace + HBM / MFMA counter passes of a bench.py workload (tools/profile_round.sh)
#   shards  <tag>                       kernel traces of the per-GPU shard shapes of the 8-GPU configs (UNet 2 x 256^2, latent 8 / 64 images)
#   wino    <tag> [variants]            fused Winograd kernels: per-layer sweep, cycle stamps of the two-tile-group kernel, TA / TD counter pass
#   probe   <tag> <name>                build + run tools/probe/<name>.hip
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
RECIPE=${1:?recipe}; TAG=${2:?tag}; shift 2
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
case "$RECIPE" in
  suite)
    ( time timeout 2400 python -m pytest tests -m gpu -q "$@" ) > "$OUT/pytest_gpu.txt" 2>&1
    tail -8 "$OUT/pytest_gpu.txt" | cut -c1-250 ;;
  bench)
    timeout 1500 python bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
    tail -c 1500 "$OUT/bench.json" ;;
  ab)
    SPEC=${1:?ENV=V,V}; shift
    VAR=${SPEC%%=*}
    for V in $(echo "${SPEC#*=}" | tr ',' ' '); do
      env IRSDE_TUNING=1 "$VAR=$V" timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-live-pmc "$@" > "$OUT/ab_${VAR}_$V.json" 2> "$OUT/ab_${VAR}_$V.err"
      echo "$VAR=$V $(grep -o '"value": *[0-9.]*' "$OUT/ab_${VAR}_$V.json" | head -1) $(grep -o '"ms_per_step": *[0-9.]*' "$OUT/ab_${VAR}_$V.json" | head -1)"
    done | tee "$OUT/ab_$VAR.txt" ;;
  profile)
    bash tools/profile_round.sh "$TAG" "$@" ;;
  shards)
    bash tools/kernel_trace.sh "$TAG/kt_unet_b2" --batch 2 > /dev/null 2>&1
    bash tools/kernel_trace.sh "$TAG/kt_latent_b8" --model latent --dtype fp16 --batch 8 > /dev/null 2>&1
    bash tools/kernel_trace.sh "$TAG/kt_latent_b64" --model latent --dtype fp16 --batch 64 > /dev/null 2>&1
    python tools/op_profile.py 2 256 0 > "$OUT/op_profile_b2.txt" 2>&1
    ls -la "$OUT" ;;
  wino)
    VARS=${1:-430,460,466}
    python tools/wino_fused64p_sweep.py "$VARS" > "$OUT/sweep.txt" 2>&1; cat "$OUT/sweep.txt"
    python tools/wino_t_stamps.py 16 465 > "$OUT/stamps.txt" 2>&1
    bash tools/pmc_ta_pass.sh "$TAG/ta" "430,460" ta > /dev/null 2>&1
    cp "$OUT/ta/pmc_ta_ta.txt" "$OUT/" 2>/dev/null; rm -rf "$OUT"/ta/q[0-9] ;;
  probe)
    NAME=${1:?probe name}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o "tools/probe/$NAME" "tools/probe/$NAME.hip" 2> "$OUT/$NAME.build.log" && timeout 600 "tools/probe/$NAME" > "$OUT/$NAME.txt" 2>&1
    tail -40 "$OUT/$NAME.txt" ;;
  *) echo "unknown recipe $RECIPE"; exit 2 ;;
esac
