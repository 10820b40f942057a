"""The benchmark workloads: BASELINE.json's configs as (scene file under scenes/, parameter overrides, description)."""
WORKLOADS = {
    # name: (scene file, params overrides, description)
    "default": ("default.txt", dict(), "configs[1]: shipped scene (Menger iterations 3), 1024x768, SSAA 3x3, 2 lights"),
    "menger2": ("menger2.txt", dict(kernel=1), "configs[0]: Menger iterations 2, 1024x768, SSAA 1, 2 lights"),
    "uhd": ("default.txt", dict(width=3840, height=2160, kernel=4), "configs[2]: shipped scene, 3840x2160, SSAA 4x4, 2 lights"),
    "closeup6": ("closeup6.txt", dict(), "configs[3]: Menger iterations 6 close-up, 1024x768, SSAA 3x3, 2 lights"),
    "synthetic_wide": ("synthetic1024_wide.txt", dict(), "configs[4] with every size in [1, 4] (the crosses' bars too): 1024 mixed primitives, 8 lights, 1024x768, SSAA 3x3"),
    "synthetic": ("synthetic1024.txt", dict(), "configs[4]: 1024 mixed primitives, 8 lights, 1024x768, SSAA 3x3"),
}
