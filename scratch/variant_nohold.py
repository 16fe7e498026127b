import sys
p=sys.argv[1]+'/prover/cache.cpp'
s=open(p).read()
s=s.replace("    zz->tb.hold.store(true, std::memory_order_release); // the deferred table build reads the base arrays: not before they are complete\n","")
open(p,'w').write(s)
