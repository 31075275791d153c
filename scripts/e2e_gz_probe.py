import json, sys
sys.path.insert(0, '.')
from topsicle_amd import e2e, synth
b,o,t = synth.make_reads(10000,15000,"CCCTAA",seed=20250920,errors=synth.ONT)
r = e2e.measure(b,o,"CCCTAA",4,6,device=0, with_cli=False) if 'with_cli' in e2e.measure.__code__.co_varnames else e2e.measure(b,o,"CCCTAA",4,6,device=0)
print(json.dumps(r["gz_file_to_results"]))
