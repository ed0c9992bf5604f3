"""Extracts the camera records of the reference's second debug dump, `check_colmap copy.md` (179 cameras printed as
`cameras.FoVx: ...`, `cameras.world_view_transform: tensor([[...]], device='cuda:0')`, ... with four decimals), into a JSON
fixture.  The fixture is DATA (FoV, image size, printed matrices), not reference source.  Run in the build container only:
    python tests/golden/extract_check_colmap_cameras.py "/root/reference/check_colmap copy.md"
"""
import json
import re
import sys


def tensor(text):
    return [float(x) for x in re.findall(r"-?\d+\.\d+(?:e[+-]?\d+)?|-?\d+\.", text)]


def parse(path, every=6):
    text = open(path).read()
    blocks = re.split(r"(?=camera\.uid: \d+)", text)
    recs = []
    for b in blocks:
        m = re.match(r"camera\.uid: (\d+)", b)
        if not m:
            continue
        g = lambda key: re.search(r"cameras\." + key + r": (.*?) \[\d\d/\d\d ", b, flags=re.S).group(1)  # noqa: E731
        wv, fp, cc = tensor(g("world_view_transform")), tensor(g("full_proj_transform")), tensor(g("camera_center"))
        assert len(wv) == 16 and len(fp) == 16 and len(cc) == 3, (m.group(1), len(wv), len(fp), len(cc))
        recs.append(dict(uid=int(m.group(1)), FoVx=float(g("FoVx")), FoVy=float(g("FoVy")), image_width=int(g("image_width")),
                         image_height=int(g("image_height")), world_view_transform=[wv[4 * i:4 * i + 4] for i in range(4)],
                         full_proj_transform=[fp[4 * i:4 * i + 4] for i in range(4)], camera_center=cc))
    return recs[::every], len(recs)


if __name__ == "__main__":
    out, n = parse(sys.argv[1])
    json.dump(out, open(__file__.replace("extract_check_colmap_cameras.py", "check_colmap_cameras.json"), "w"), indent=1)
    print(len(out), "of", n, "cameras")
