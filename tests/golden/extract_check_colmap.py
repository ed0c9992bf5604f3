"""Extracts a few keyframe records from the reference's own debug dump check_colmap.md (output of
GaussianKeyframe::logger, src/gaussian_keyframe.cpp:293-302) into a small JSON fixture.  The fixture is
DATA (printed 4x4 matrices, FoV, image size), not reference source.  Run in the build container only:
    python tests/golden/extract_check_colmap.py /root/reference/check_colmap.md
"""
import json
import re
import sys


def parse(path, limit=8):
    text = open(path).read()
    recs = []
    for m in re.finditer(r"\[GaussianKeyframe\]fid_: (\d+), camera_id_ = (\d+), FoVx_ = ([\d.eE+-]+), FoVy_ = ([\d.eE+-]+), "
                         r"image_width_ = (\d+), image_height_ = (\d+), world_view_transform_ = (.*?)\[ CUDAFloatType\{4,4\} \], "
                         r"projection_matrix_ = (.*?)\[ CUDAFloatType\{4,4\} \], full_proj_transform_ = (.*?)\[ CUDAFloatType\{4,4\} \], "
                         r"camera_center_ = (.*?)\[ CUDAFloatType\{3\} \]", text, flags=re.S):
        nums = lambda s: [float(x) for x in s.split()]  # noqa: E731
        mat = lambda s: [nums(s)[4 * i:4 * i + 4] for i in range(4)]  # noqa: E731
        recs.append(dict(fid=int(m.group(1)), FoVx=float(m.group(3)), FoVy=float(m.group(4)), image_width=int(m.group(5)),
                         image_height=int(m.group(6)), world_view_transform=mat(m.group(7)), projection_matrix=mat(m.group(8)),
                         full_proj_transform=mat(m.group(9)), camera_center=nums(m.group(10))))
        if len(recs) >= limit:
            break
    return recs


if __name__ == "__main__":
    out = parse(sys.argv[1])
    json.dump(out, open(__file__.replace("extract_check_colmap.py", "check_colmap_keyframes.json"), "w"), indent=1)
    print(len(out), "keyframes")
