cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2; do for q in 0 1 2 3; do echo -n "qsplit=$q "; SSG_BWD_QSPLIT=$q python bench.py --no-extra --no-cpu-baseline --no-module --steps 100 --warmup 20 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('c2 %.4f bwd_dense %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms']['ssg_bwd_dense<25,9,3>']))"; done; done
