set -e
cd $GRAFT_REPO_ROOT
T=/tmp/se; mkdir -p $T
for f in idx.bwt idx.sa idx.pac idx.ann idx.amb; do cp tests/golden/var/$f $T/; done
zcat tests/golden/var/r1.fq.gz > $T/r1.fq; zcat tests/golden/var/r2.fq.gz > $T/r2.fq
zcat tests/golden/var/ref.ksw2.sam.gz > $T/ref.sam; zcat tests/golden/var/ref.vcf.default.gz > $T/ref.vcf
PYTHONPATH=$GRAFT_REPO_ROOT python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 -m mapcaller_amd.run -backend gloo -i $T/idx -f $T/r1.fq -f2 $T/r2.fq -alg ksw2 -sam $T/o.sam -vcf $T/o.vcf -batch 2000 > /dev/null 2>&1
echo "SAM lines differing from the single-stream reference: $(diff $T/ref.sam $T/o.sam | grep -c '^<') of $(wc -l < $T/ref.sam)"
echo "VCF body lines differing: $(diff <(grep -v '^##command_line\|^##reference' $T/ref.vcf) <(grep -v '^##command_line\|^##reference' $T/o.vcf) | grep -c '^[<>]')"
