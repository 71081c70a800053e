// Headless stand-in for the reference's viewport.cpp (GLUT/OpenGL UI, out of scope:
// SURVEY.md §2 row 12).  It only gives bodies to the UI-side virtuals that the
// reference declares in its plugin headers and defines in viewport.cpp:545-685,
// so that the reference's renderer TUs link without freeglut.  No renderer
// behaviour lives here.  Test infrastructure only (oracle/_ref build).
#include "Scenes/scene.h"
#include "Objects/objects.h"
#include "Lights/lights.h"
#include "Materials/materials.h"
#include "Textures/texture.h"

void BeginRender();
void ShowViewport() { BeginRender(); }
void Sphere::ViewportDisplay(const Material *) const {}
void Plane::ViewportDisplay(const Material *) const {}
void TriObj::ViewportDisplay(const Material *) const {}
void MtlBlinn::SetViewportMaterial(int) const {}
void GenLight::SetViewportParam(int, ColorA, ColorA, Vec4f) const {}
void PointLight::SetViewportLight(int) const {}
bool TextureFile::SetViewportTexture() const { return false; }
bool TextureChecker::SetViewportTexture() const { return false; }
