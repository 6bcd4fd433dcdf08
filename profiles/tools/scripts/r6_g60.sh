cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_ptf_hip.py tests/test_composed_dropin.py tests/test_configs_4_5.py tests/test_pipeline_c1.py -m gpu -q 2>&1 | tail -2
for sp in 1 0; do
echo "== FS_PTF_WS_SPLIT=$sp"
for w in ptf_3_views ptf_10_views; do
rm -rf /tmp/prof_x
FS_PTF_WS_SPLIT=$sp rocprofv3 --kernel-trace -d /tmp/prof_x -o x --output-format csv -- python profiles/tools/fwd_traffic.py run $w 6 > /dev/null 2>&1
python profiles/tools/kstats.py /tmp/prof_x | grep "write_state" | sed "s/^/$w /"
done
FS_PTF_WS_SPLIT=$sp python - <<'PY'
import torch, bench_encoder as be
dev = torch.device("cuda:0")
for V, hw in ((3, (968, 1296)), (10, None), (30, None)):
    try:
        r = be.bench_ptf(dev, 12, 2, V=V, **({"h": hw[0], "w": hw[1]} if hw else {}), train=False) if False else None
    except Exception as e:
        print(e)
PY
done
