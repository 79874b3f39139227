from .camera import Camera, OrbitCamera, Renderer
