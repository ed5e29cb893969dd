O=$GRAFT_REPO_ROOT/gpurun_out/r3r; mkdir -p $O; cd $GRAFT_REPO_ROOT
python3 tools/epoch_profile.py ppi --spans > $O/ppi_spans.log 2>&1
cd /tmp && export TMPDIR=/tmp
for c in pubmed ppi; do timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/${c}_prof -- python3 $GRAFT_REPO_ROOT/tools/epoch_profile.py $c --epochs 30 > $O/${c}_prof.log 2>&1; done
cd $GRAFT_REPO_ROOT
for c in pubmed ppi; do python3 tools/epoch_sequence.py $O/${c}_prof --epochs 35 > $O/${c}_seq.txt 2>&1; rm -rf $O/${c}_prof; done
