#!/bin/bash
# usage: tools/kres.sh file.hip [extra hipcc flags] -- compact per-kernel resource usage (VGPR / scratch / LDS / occupancy)
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -c "$f" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
 awk '/Function Name/{n=$0; sub(/.*Function Name: /,"",n); sub(/ \[.*/,"",n)}
      /VGPRs:/{v=$0; sub(/.* VGPRs: /,"",v); sub(/ \[.*/,"",v)}
      /ScratchSize/{s=$0; sub(/.*: /,"",s); sub(/ \[.*/,"",s)}
      /Occupancy/{o=$0; sub(/.*: /,"",o); sub(/ \[.*/,"",o)}
      /LDS Size/{l=$0; sub(/.*: /,"",l); sub(/ \[.*/,"",l); printf "%-90s vgpr=%s scratch=%s occ=%s lds=%s\n", substr(n,1,90), v, s, o, l}'
