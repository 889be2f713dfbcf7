"""No-op stand-in for the reference renderer (/root/reference/code/engine/render_engine.py:246-280):
visualisation is out of scope; the trajopt scripts keep calling the same methods."""


class Renderer:
    def __init__(self, sys, name="scene", option="None"):
        self.sys = sys
        self.name = name
        self.option = option
        self.save_dir = None

    def set_save_dir(self, path):
        import os
        self.save_dir = path
        os.makedirs(path, exist_ok=True)

    def render(self, frame):
        pass

    def end_rendering(self, it):
        pass
