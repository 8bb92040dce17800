"""Parse the reference's shipped YAML configs (config/{uncond,text,rearrange}/*.yaml) into ONE JSON fixture so that the
drop-in tests can drive ``build_network`` / ``train_on_batch`` with exactly the keys and values a user of the reference has
(the GPU box has no /root/reference).  Run in the build container:  python tests/golden/make_config_fixture.py"""
import glob
import json
import os

import yaml

REF = "/root/reference/config"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_configs.json")


def main():
    cfgs = {}
    for sub in ("uncond", "text", "rearrange"):
        for path in sorted(glob.glob(os.path.join(REF, sub, "*.yaml"))):
            with open(path) as f:
                cfgs["%s/%s" % (sub, os.path.basename(path))] = yaml.load(f, Loader=yaml.Loader)
    with open(OUT, "w") as f:
        json.dump(cfgs, f, indent=1, sort_keys=True)
    print("wrote %d configs to %s" % (len(cfgs), OUT))


if __name__ == "__main__":
    main()
